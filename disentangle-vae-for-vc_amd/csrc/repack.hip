// Weight-derived operand layouts, refreshed by ONE launch per training step.
//
// The contraction kernels want every operand k-contiguous (gemm.hip) and the LSTM frame kernels want W_hh in MFMA
// fragment order (lstm.hip).  Those layouts are functions of the weights alone, and the weights change once per step
// (in Adam), so they are produced once per step — by one launch that walks a table of descriptors — instead of by one
// small launch per use (round 1: 42 pack / transpose launches + 19 ATen bias adds per step):
//   CONV_T     Wp[5][Cout][Cin] (the layout conv weights live in, see model/disentangled_vae.py) -> Wpt[5][Cin][Cout]
//              for the data gradient;
//   LSTM_PACK  W_hh[4H][H] -> forward / backward fragment packs (fp32 for v_mfma_f32_16x16x4_f32, bf16 or three bf16
//              planes for v_mfma_f32_16x16x32_bf16, per pack as the descriptor says);
//   TRANSPOSE  W[R][C] -> W^T[C][R]   (W_ih^T for the input-projection data gradients, W_hh^T for the H = 64 and
//              generic backward recurrences);
//   ADD2       b_ih + b_hh (nn.LSTM keeps two bias vectors; the kernels add one);
//   COPY_F32   fp32 copies (W_ih of the two directions of a BiLSTM layer side by side: one input projection for both);
//   CAST_BF16  bf16 copies of weights (bf16 compute mode: the B operands of the forward / data-gradient contractions);
//              CONV_T and TRANSPOSE can write bf16 as well.
// All of it is HBM-bound byte shuffling: 32x32 tiles through LDS so both sides move whole 128-B lines.
#include "common.h"

namespace {

constexpr int MAX_DESC = 72;   // the table travels as a kernel argument (3 756 bytes <= 4 KB)

struct Desc {
  int kind, d0, d1, d2;
  const float* src;
  const float* src2;
  float* dst;
  float* dst2;
};
struct Table {
  int n, pad_;
  int blk_begin[MAX_DESC + 1];
  Desc d[MAX_DESC];
};

// [R][C] -> [C][R], tiles lb, lb + nb, ...; out16: the destination is bf16
__device__ __forceinline__ void transpose_tiles(const float* __restrict__ in, float* __restrict__ out, int R, int C,
                                                int lb, int nb, float (*tile)[33], bool out16 = false, int ldo = 0) {
  if (ldo == 0) ldo = R;      // (ldo > R: the transposes of several sources side by side in one destination)
  const int tc = (C + 31) / 32, tr = (R + 31) / 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int t = lb; t < tc * tr; t += nb) {
    const int c0 = (t % tc) * 32, r0 = (t / tc) * 32;
    __syncthreads();
    for (int k = ty; k < 32; k += 8) {
      const int r = r0 + k, c = c0 + tx;
      tile[k][tx] = (r < R && c < C) ? in[(int64_t)r * C + c] : 0.f;
    }
    __syncthreads();
    for (int k = ty; k < 32; k += 8) {
      const int c = c0 + k, r = r0 + tx;
      if (c < C && r < R) {
        if (out16) reinterpret_cast<__bf16*>(out)[(int64_t)c * ldo + r] = (__bf16)tile[tx][k];
        else out[(int64_t)c * ldo + r] = tile[tx][k];
      }
    }
  }
}

// W_hh [4H,H] -> fragment-packed copies.  fwd: [(g*n_j+jb)][kc][lane][4] <- W[g*H + jb*16 + r][kc*16 + 4q + e];
// bwd: [(jb*4+w)][kc][lane][4] <- W[w*H + kc*16 + 4q + e][jb*16 + r]   (lane = q*16 + r)
__device__ __forceinline__ void lstm_pack_f32(const float* __restrict__ W, float* __restrict__ pf, float* __restrict__ pb,
                                              int H, int64_t first, int64_t stride) {
  const int n_j = H / 16, nkc = H / 16;
  const int64_t total = (int64_t)4 * n_j * nkc * 64;
  for (int64_t i = first; i < total; i += stride) {
    const int lane = (int)(i & 63);
    const int64_t c = i >> 6;
    const int kc = (int)(c % nkc);
    const int64_t gj = c / nkc;
    const int r = lane & 15, q = lane >> 4;
    if (pf) {
      const int g = (int)(gj / n_j), jb = (int)(gj % n_j);
      reinterpret_cast<f32x4*>(pf)[i] =
          *reinterpret_cast<const f32x4*>(W + ((int64_t)g * H + jb * 16 + r) * H + kc * 16 + 4 * q);
    }
    if (pb) {
      const int jb = (int)(gj / 4), w = (int)(gj % 4);
      f32x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = W[((int64_t)w * H + kc * 16 + 4 * q + e) * H + jb * 16 + r];
      reinterpret_cast<f32x4*>(pb)[i] = v;
    }
  }
}

// bf16 fragment packing (v_mfma_f32_16x16x32_bf16): 32-deep chunks, lane (r, q) holds k = 32kc + 8q + j, j = 0..7.
// fwd: [(g*n_j+jb)][kc][lane][8] <- W[g*H + jb*16 + r][32kc + 8q + j];  bwd: [(jb*4+w)][kc][lane][8] <- W[w*H + 32kc + 8q + j][jb*16 + r]
__device__ __forceinline__ void lstm_pack_bf16(const float* __restrict__ W, __bf16* __restrict__ pf,
                                               __bf16* __restrict__ pb, int H, int64_t first, int64_t stride) {
  const int n_j = H / 16, nkc = H / 32;
  const int64_t total = (int64_t)4 * n_j * nkc * 64;
  for (int64_t i = first; i < total; i += stride) {
    const int lane = (int)(i & 63);
    const int64_t c = i >> 6;
    const int kc = (int)(c % nkc);
    const int64_t gj = c / nkc;
    const int r = lane & 15, q = lane >> 4;
    if (pf) {
      const int g = (int)(gj / n_j), jb = (int)(gj % n_j);
      const float* src = W + ((int64_t)g * H + jb * 16 + r) * H + kc * 32 + 8 * q;
      bf16x8 v;
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = (__bf16)src[e];
      reinterpret_cast<bf16x8*>(pf)[i] = v;
    }
    if (pb) {
      const int jb = (int)(gj / 4), w = (int)(gj % 4);
      bf16x8 v;
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = (__bf16)W[((int64_t)w * H + kc * 32 + 8 * q + e) * H + jb * 16 + r];
      reinterpret_cast<bf16x8*>(pb)[i] = v;
    }
  }
}

// fp32x3 fragment packing: as the bf16 packing with THREE planes per (chunk, lane): w = p0 + p1 + p2 exactly (two
// round-to-nearest splits, the last residual is exact in bf16).  [..][kc][plane][lane][8]
__device__ __forceinline__ void split3_scalar(float x, __bf16& a, __bf16& b, __bf16& c) {
  a = (__bf16)x;
  const float r1 = x - (float)a;
  b = (__bf16)r1;
  c = (__bf16)(r1 - (float)b);
}
__device__ __forceinline__ void lstm_pack_x3(const float* __restrict__ W, __bf16* __restrict__ pf,
                                             __bf16* __restrict__ pb, int H, int64_t first, int64_t stride) {
  const int n_j = H / 16, nkc = H / 32;
  const int64_t total = (int64_t)4 * n_j * nkc * 64;
  for (int64_t i = first; i < total; i += stride) {
    const int lane = (int)(i & 63);
    const int64_t c = i >> 6;                       // (group, chunk)
    const int kc = (int)(c % nkc);
    const int64_t gj = c / nkc;
    const int r = lane & 15, q = lane >> 4;
    if (pf) {
      const int g = (int)(gj / n_j), jb = (int)(gj % n_j);
      const float* src = W + ((int64_t)g * H + jb * 16 + r) * H + kc * 32 + 8 * q;
      bf16x8 v[3];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        __bf16 p0, p1, p2;
        split3_scalar(src[e], p0, p1, p2);
        v[0][e] = p0; v[1][e] = p1; v[2][e] = p2;
      }
#pragma unroll
      for (int p = 0; p < 3; ++p) reinterpret_cast<bf16x8*>(pf)[(c * 3 + p) * 64 + lane] = v[p];
    }
    if (pb) {
      const int jb = (int)(gj / 4), w = (int)(gj % 4);
      bf16x8 v[3];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        __bf16 p0, p1, p2;
        split3_scalar(W[((int64_t)w * H + kc * 32 + 8 * q + e) * H + jb * 16 + r], p0, p1, p2);
        v[0][e] = p0; v[1][e] = p1; v[2][e] = p2;
      }
#pragma unroll
      for (int p = 0; p < 3; ++p) reinterpret_cast<bf16x8*>(pb)[(c * 3 + p) * 64 + lane] = v[p];
    }
  }
}

__global__ __launch_bounds__(256) void lstm_pack_w_x3_kernel(const float* __restrict__ W, __bf16* __restrict__ pf,
                                                             __bf16* __restrict__ pb, int H) {
  lstm_pack_x3(W, pf, pb, H, (int64_t)blockIdx.x * 256 + threadIdx.x, (int64_t)gridDim.x * 256);
}
__global__ __launch_bounds__(256) void lstm_pack_w_kernel(const float* __restrict__ W, float* __restrict__ pf,
                                                          float* __restrict__ pb, int H) {
  lstm_pack_f32(W, pf, pb, H, (int64_t)blockIdx.x * 256 + threadIdx.x, (int64_t)gridDim.x * 256);
}
__global__ __launch_bounds__(256) void lstm_pack_w_bf16_kernel(const float* __restrict__ W, __bf16* __restrict__ pf,
                                                               __bf16* __restrict__ pb, int H) {
  lstm_pack_bf16(W, pf, pb, H, (int64_t)blockIdx.x * 256 + threadIdx.x, (int64_t)gridDim.x * 256);
}

__global__ __launch_bounds__(256) void repack_all_kernel(const Table t) {
  __shared__ float tile[32][33];
  int di = 0;
  while (di + 1 < t.n && (int)blockIdx.x >= t.blk_begin[di + 1]) ++di;   // uniform scan, n <= 56
  const Desc& d = t.d[di];
  const int lb = blockIdx.x - t.blk_begin[di], nb = t.blk_begin[di + 1] - t.blk_begin[di];
  switch (d.kind) {
    case DVAE_REPACK_CONV_T: {   // d0 = Cout, d1 = Cin: five [Cout][Cin] -> [Cin][Cout] transposes
      const int64_t cc = (int64_t)d.d0 * d.d1;
      for (int tap = 0; tap < 5; ++tap) {
        float* dst = d.d2 ? (float*)((__bf16*)d.dst + tap * cc) : d.dst + tap * cc;
        transpose_tiles(d.src + tap * cc, dst, d.d0, d.d1, lb, nb, tile, d.d2 != 0);
      }
      break;
    }
    case DVAE_REPACK_LSTM_PACK: {   // d0 = H, d1 / d2 = DVAE_MODE_* of the forward / backward pack
      const int64_t first = (int64_t)lb * 256 + threadIdx.x, stride = (int64_t)nb * 256;
      for (int which = 0; which < 2; ++which) {
        float* dst = which ? d.dst2 : d.dst;
        const int m = which ? d.d2 : d.d1;
        if (!dst) continue;
        float* pf = which ? nullptr : dst;
        float* pb = which ? dst : nullptr;
        if (m == DVAE_MODE_BF16) lstm_pack_bf16(d.src, (__bf16*)pf, (__bf16*)pb, d.d0, first, stride);
        else if (m == DVAE_MODE_F32X3) lstm_pack_x3(d.src, (__bf16*)pf, (__bf16*)pb, d.d0, first, stride);
        else lstm_pack_f32(d.src, pf, pb, d.d0, first, stride);
      }
      break;
    }
    case DVAE_REPACK_TRANSPOSE:   // d0 = R, d1 = C, d2 & 1: bf16 destination, d2 >> 1: row stride of the destination
      transpose_tiles(d.src, d.dst, d.d0, d.d1, lb, nb, tile, (d.d2 & 1) != 0, d.d2 >> 1);
      break;
    case DVAE_REPACK_COPY_F32: {  // n = d0 elements, a multiple of 4
      const int64_t n4 = (int64_t)d.d0 >> 2;
      for (int64_t i = (int64_t)lb * 256 + threadIdx.x; i < n4; i += (int64_t)nb * 256)
        reinterpret_cast<f32x4*>(d.dst)[i] = reinterpret_cast<const f32x4*>(d.src)[i];
      break;
    }
    case DVAE_REPACK_CAST_BF16: { // n = d0 * d1 elements (d1 >= 1), a multiple of 4
      const int64_t n4 = ((int64_t)d.d0 * d.d1) >> 2;
      for (int64_t i = (int64_t)lb * 256 + threadIdx.x; i < n4; i += (int64_t)nb * 256)
        reinterpret_cast<bf16x4*>(d.dst)[i] = __builtin_convertvector(reinterpret_cast<const f32x4*>(d.src)[i], bf16x4);
      break;
    }
    case DVAE_REPACK_ADD2:        // d0 = n
      for (int i = lb * 256 + threadIdx.x; i < d.d0; i += nb * 256) d.dst[i] = d.src[i] + d.src2[i];
      break;
    default:
      break;
  }
}

}  // namespace

DVAE_API int dvae_lstm_pack_w(const float* w_hh, float* packed_fwd, float* packed_bwd, int H, void* stream) {
  if (!w_hh || (!packed_fwd && !packed_bwd) || H < 64 || (H & 63)) return DVAE_EINVAL;
  const int64_t total = (int64_t)4 * (H / 16) * (H / 16) * 64;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(lstm_pack_w_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w_hh, packed_fwd, packed_bwd, H);
  return dvae_check_launch();
}

DVAE_API int dvae_lstm_pack_w_bf16(const float* w_hh, void* packed_fwd, void* packed_bwd, int H, void* stream) {
  if (!w_hh || (!packed_fwd && !packed_bwd) || H < 512 || (H % 512)) return DVAE_EINVAL;
  const int64_t total = (int64_t)4 * (H / 16) * (H / 32) * 64;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(lstm_pack_w_bf16_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w_hh,
                     (__bf16*)packed_fwd, (__bf16*)packed_bwd, H);
  return dvae_check_launch();
}

DVAE_API int dvae_lstm_pack_w_x3(const float* w_hh, void* packed_fwd, void* packed_bwd, int H, void* stream) {
  if (!w_hh || (!packed_fwd && !packed_bwd) || H < 512 || (H % 512)) return DVAE_EINVAL;
  const int64_t total = (int64_t)4 * (H / 16) * (H / 32) * 64;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(lstm_pack_w_x3_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w_hh,
                     (__bf16*)packed_fwd, (__bf16*)packed_bwd, H);
  return dvae_check_launch();
}

DVAE_API int dvae_repack_all(const dvae_repack_desc_t* descs, int n, void* stream) {
  if (!descs || n < 1 || n > MAX_DESC) return DVAE_EINVAL;
  Table t{};
  t.n = n;
  int blocks = 0;
  for (int i = 0; i < n; ++i) {
    const dvae_repack_desc_t& s = descs[i];
    int64_t elems = 0;
    switch (s.kind) {
      case DVAE_REPACK_CONV_T:
        if (!s.src || !s.dst || s.d0 < 1 || s.d1 < 1) return DVAE_EINVAL;
        elems = (int64_t)5 * s.d0 * s.d1;
        break;
      case DVAE_REPACK_LSTM_PACK:
        if (!s.src || (!s.dst && !s.dst2) || s.d0 < 64 || (s.d0 & 63)) return DVAE_EINVAL;
        for (int m : {s.d1, s.d2})
          if ((m != DVAE_MODE_F32 && m != DVAE_MODE_BF16 && m != DVAE_MODE_F32X3) || (m != DVAE_MODE_F32 && (s.d0 % 512)))
            return DVAE_EINVAL;
        elems = (int64_t)4 * s.d0 * s.d0;
        break;
      case DVAE_REPACK_TRANSPOSE:
        if (!s.src || !s.dst || s.d0 < 1 || s.d1 < 1 || s.d2 < 0 || ((s.d2 >> 1) && (s.d2 >> 1) < s.d0)) return DVAE_EINVAL;
        elems = (int64_t)s.d0 * s.d1;
        break;
      case DVAE_REPACK_COPY_F32:
        if (!s.src || !s.dst || s.d0 < 4 || (s.d0 & 3) || ((((uintptr_t)s.src) | ((uintptr_t)s.dst)) & 15)) return DVAE_EINVAL;
        elems = s.d0;
        break;
      case DVAE_REPACK_ADD2:
        if (!s.src || !s.src2 || !s.dst || s.d0 < 1) return DVAE_EINVAL;
        elems = s.d0;
        break;
      case DVAE_REPACK_CAST_BF16:
        if (!s.src || !s.dst || s.d0 < 1 || s.d1 < 1 || (((int64_t)s.d0 * s.d1) & 3)) return DVAE_EINVAL;
        elems = (int64_t)s.d0 * s.d1;
        break;
      default:
        return DVAE_EINVAL;
    }
    t.d[i] = Desc{s.kind, s.d0, s.d1, s.d2, (const float*)s.src, (const float*)s.src2, (float*)s.dst, (float*)s.dst2};
    int64_t nb = (elems + 8191) / 8192;     // ~8 K elements per workgroup
    if (nb < 1) nb = 1;
    if (nb > 512) nb = 512;
    t.blk_begin[i] = blocks;
    blocks += (int)nb;
  }
  t.blk_begin[n] = blocks;
  hipLaunchKernelGGL(repack_all_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, t);
  return dvae_check_launch();
}
