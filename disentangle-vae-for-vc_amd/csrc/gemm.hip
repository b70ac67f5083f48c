// fp32 dense contraction on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32).
//
// One kernel serves every GEMM-shaped piece of the training step: Linear fwd/bwd, LSTM input
// projections and weight gradients, and the k=5 Conv1d as an implicit GEMM over frame-major rows
// (a tap is a row shift of +-N*(tap-2), so padding is a plain range check on the row index).
//
// Tile: 128x128x16 per 256-thread workgroup (4 waves as 2x2, each wave 2x2 MFMA tiles of 32x32,
// 64 accumulator VGPRs).  Operands are staged global -> registers -> LDS in a k-major image
// [BK][128+pad] so that one ds_read_b32 per lane feeds an MFMA operand: lanes 0-31 read 32
// consecutive floats of k-row 2s, lanes 32-63 of k-row 2s+1 (the A[i][k]/B[k][j] lane map of
// the 32x32x2 instruction) — conflict-free.  fp32 MFMA issues once per 64 cycles per SIMD, so
// LDS bandwidth is far from critical and the structure stays simple: register prefetch of the
// next tile, two LDS buffers, one barrier per k-tile.
#include "common.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 16, NTHR = 256;

struct GemmParams {
  const float* A;
  const float* B;
  float* C;
  const float* bias;
  int M, N, K;
  int64_t lda, ldb, ldc;
  int taps;              // 1 or 5
  int tap_mode;          // 0 none | 1 loop over taps, one output | 2 one output per tap (grid.z)
  int64_t a_row_shift;   // mode 1: A row offset per (tap-2)
  int64_t b_tap_stride;  // mode 1: elements between per-tap B matrices
  int64_t bk_row_shift;  // mode 2: B k-row offset per (tap-2)
  int64_t c_tap_stride;  // mode 2: elements between per-tap C matrices
  int split_k;
  int k_per_split;       // multiple of BK
  int act, epi;
  int tiles_m;
};

template <bool A_KC, bool B_KC>
__global__ __launch_bounds__(NTHR) void gemm_f32_kernel(const GemmParams p) {
  constexpr int LDA = A_KC ? 129 : 132;
  constexpr int LDB = B_KC ? 129 : 132;
  __shared__ __attribute__((aligned(16))) float As[2][BK * LDA];
  __shared__ __attribute__((aligned(16))) float Bs[2][BK * LDB];

  const int t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, kh = lane >> 5;

  const int tile_m = blockIdx.x % p.tiles_m;
  const int tile_n = blockIdx.x / p.tiles_m;
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  int tap_fixed = 0, ks = blockIdx.z;
  if (p.tap_mode == 2) {
    tap_fixed = blockIdx.z / p.split_k;
    ks = blockIdx.z % p.split_k;
  }
  const int k_begin = ks * p.k_per_split;
  const int k_end = min(p.K, k_begin + p.k_per_split);
  const int kiters = (k_end - k_begin + BK - 1) / BK;
  const int n_iters = (p.tap_mode == 1 ? p.taps : 1) * kiters;

  float* __restrict__ C = p.C + (p.tap_mode == 2 ? (int64_t)tap_fixed * p.c_tap_stride : 0);

  f32x4 ra[2], rb[2];

  auto load_tiles = [&](int it) {
    int tap = 0, kit = it;
    if (p.tap_mode == 1) {
      tap = it / kiters;
      kit = it - tap * kiters;
    }
    const int k0 = k_begin + kit * BK;
    // ---- A ----
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int idx = t + NTHR * j;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (A_KC) {
        const int row = idx >> 2, kq = idx & 3;
        const int k = k0 + 4 * kq;
        int64_t m = m0 + row;
        bool ok = (m < p.M) && (k < k_end);
        if (p.tap_mode == 1) {
          m += (int64_t)(tap - 2) * p.a_row_shift;
          ok = ok && (m >= 0) && (m < p.M);
        }
        if (ok) v = *reinterpret_cast<const f32x4*>(p.A + m * p.lda + k);
      } else {
        const int kr = idx >> 5, m4 = idx & 31;
        const int k = k0 + kr;
        const int m = m0 + 4 * m4;
        if (k < k_end && m < p.M) v = *reinterpret_cast<const f32x4*>(p.A + (int64_t)k * p.lda + m);
      }
      ra[j] = v;
    }
    // ---- B ----
    const float* __restrict__ Bp = p.B + (p.tap_mode == 1 ? (int64_t)tap * p.b_tap_stride : 0);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int idx = t + NTHR * j;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (B_KC) {
        const int row = idx >> 2, kq = idx & 3;
        const int k = k0 + 4 * kq;
        const int n = n0 + row;
        if (n < p.N && k < k_end) v = *reinterpret_cast<const f32x4*>(Bp + (int64_t)n * p.ldb + k);
      } else {
        const int kr = idx >> 5, n4 = idx & 31;
        int64_t k = k0 + kr;
        const int n = n0 + 4 * n4;
        bool ok = (k < k_end) && (n < p.N);
        if (p.tap_mode == 2) {
          k += (int64_t)(tap_fixed - 2) * p.bk_row_shift;
          ok = ok && (k >= 0) && (k < p.K);
        }
        if (ok) v = *reinterpret_cast<const f32x4*>(Bp + k * p.ldb + n);
      }
      rb[j] = v;
    }
  };

  auto store_tiles = [&](int buf) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int idx = t + NTHR * j;
      if (A_KC) {
        const int row = idx >> 2, kq = idx & 3;
        float* d = &As[buf][(4 * kq) * LDA + row];
        d[0] = ra[j][0];
        d[LDA] = ra[j][1];
        d[2 * LDA] = ra[j][2];
        d[3 * LDA] = ra[j][3];
      } else {
        const int kr = idx >> 5, m4 = idx & 31;
        *reinterpret_cast<f32x4*>(&As[buf][kr * LDA + 4 * m4]) = ra[j];
      }
      if (B_KC) {
        const int row = idx >> 2, kq = idx & 3;
        float* d = &Bs[buf][(4 * kq) * LDB + row];
        d[0] = rb[j][0];
        d[LDB] = rb[j][1];
        d[2 * LDB] = rb[j][2];
        d[3 * LDB] = rb[j][3];
      } else {
        const int kr = idx >> 5, n4 = idx & 31;
        *reinterpret_cast<f32x4*>(&Bs[buf][kr * LDB + 4 * n4]) = rb[j];
      }
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  if (n_iters > 0) {
    load_tiles(0);
    store_tiles(0);
  }
  __syncthreads();

  int cur = 0;
  for (int it = 0; it < n_iters; ++it) {
    const bool more = (it + 1 < n_iters);
    if (more) load_tiles(it + 1);
    const float* __restrict__ as = &As[cur][kh * LDA + wm * 64 + l31];
    const float* __restrict__ bs = &Bs[cur][kh * LDB + wn * 64 + l31];
#pragma unroll
    for (int s = 0; s < BK / 2; ++s) {
      const float a0 = as[(2 * s) * LDA], a1 = as[(2 * s) * LDA + 32];
      const float b0 = bs[(2 * s) * LDB], b1 = bs[(2 * s) * LDB + 32];
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
    }
    if (more) store_tiles(cur ^ 1);
    __syncthreads();
    cur ^= 1;
  }

  // ---- epilogue: C/D lane map of the 32x32 tile: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
  const bool add_bias = (p.bias != nullptr) && (ks == 0);
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const int col = n0 + wn * 64 + nt * 32 + l31;
      if (col >= p.N) continue;
      const float bv = add_bias ? p.bias[col] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * 64 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
        if (row >= p.M) continue;
        float v = acc[mt][nt][r] + bv;
        float* c = C + (int64_t)row * p.ldc + col;
        if (p.epi == DVAE_EPI_STORE) {
          *c = act_apply(v, p.act);
        } else if (p.epi == DVAE_EPI_ACCUM) {
          *c += v;
        } else {
          atomicAdd(c, v);
        }
      }
    }
  }
}

int launch_gemm(GemmParams& p, bool a_kc, bool b_kc, hipStream_t s) {
  if (p.M <= 0 || p.N <= 0 || p.K <= 0) return DVAE_EINVAL;
  if (!p.A || !p.B || !p.C) return DVAE_EINVAL;
  if ((p.lda & 3) || (p.ldb & 3)) return DVAE_EINVAL;
  if ((a_kc || b_kc) && (p.K & 3)) return DVAE_EINVAL;
  if ((((uintptr_t)p.A) | ((uintptr_t)p.B)) & 15) return DVAE_EINVAL;
  if (!a_kc && (p.M & 3)) return DVAE_EINVAL;
  if (!b_kc && (p.N & 3)) return DVAE_EINVAL;
  if (p.split_k < 1) p.split_k = 1;
  if (p.split_k > 1 && (p.epi != DVAE_EPI_ATOMIC || p.act != DVAE_ACT_NONE)) return DVAE_EINVAL;
  if (p.epi != DVAE_EPI_STORE && p.act != DVAE_ACT_NONE) return DVAE_EINVAL;
  int kps = (p.K + p.split_k - 1) / p.split_k;
  kps = ((kps + BK - 1) / BK) * BK;
  p.k_per_split = kps;
  p.split_k = (p.K + kps - 1) / kps;
  p.tiles_m = (p.M + BM - 1) / BM;
  const int tiles_n = (p.N + BN - 1) / BN;
  dim3 grid(p.tiles_m * tiles_n, 1, p.split_k * (p.tap_mode == 2 ? p.taps : 1));
  dim3 block(NTHR);
  ProfScope prof(1, s, 2.0 * p.M * p.N * (double)p.K * (p.tap_mode ? p.taps : 1));
  if (a_kc && b_kc)
    hipLaunchKernelGGL((gemm_f32_kernel<true, true>), grid, block, 0, s, p);
  else if (a_kc && !b_kc)
    hipLaunchKernelGGL((gemm_f32_kernel<true, false>), grid, block, 0, s, p);
  else if (!a_kc && b_kc)
    hipLaunchKernelGGL((gemm_f32_kernel<false, true>), grid, block, 0, s, p);
  else
    hipLaunchKernelGGL((gemm_f32_kernel<false, false>), grid, block, 0, s, p);
  return dvae_check_launch();
}

}  // namespace

DVAE_API int dvae_gemm_f32(const float* A, const float* B, float* C, const float* bias, int M, int N, int K,
                           int64_t lda, int64_t ldb, int64_t ldc, int a_kcontig, int b_kcontig, int act,
                           int epi, int split_k, void* stream) {
  GemmParams p{};
  p.A = A; p.B = B; p.C = C; p.bias = bias;
  p.M = M; p.N = N; p.K = K;
  p.lda = lda; p.ldb = ldb; p.ldc = ldc;
  p.taps = 1; p.tap_mode = 0;
  p.split_k = split_k; p.act = act; p.epi = epi;
  return launch_gemm(p, a_kcontig != 0, b_kcontig != 0, (hipStream_t)stream);
}

DVAE_API int dvae_conv5_fwd(const float* X, const float* Wp, const float* bias, float* Y, int R, int N,
                            int Cin, int Cout, void* stream) {
  GemmParams p{};
  p.A = X; p.B = Wp; p.C = Y; p.bias = bias;
  p.M = R; p.N = Cout; p.K = Cin;
  p.lda = Cin; p.ldb = Cin; p.ldc = Cout;
  p.taps = 5; p.tap_mode = 1;
  p.a_row_shift = N; p.b_tap_stride = (int64_t)Cout * Cin;
  p.split_k = 1; p.act = DVAE_ACT_NONE; p.epi = DVAE_EPI_STORE;
  return launch_gemm(p, true, true, (hipStream_t)stream);
}

DVAE_API int dvae_conv5_dgrad(const float* dY, const float* Wp, float* dX, int R, int N, int Cin, int Cout,
                              void* stream) {
  GemmParams p{};
  p.A = dY; p.B = Wp; p.C = dX; p.bias = nullptr;
  p.M = R; p.N = Cin; p.K = Cout;
  p.lda = Cout; p.ldb = Cin; p.ldc = Cin;
  p.taps = 5; p.tap_mode = 1;
  p.a_row_shift = -(int64_t)N; p.b_tap_stride = (int64_t)Cout * Cin;
  p.split_k = 1; p.act = DVAE_ACT_NONE; p.epi = DVAE_EPI_STORE;
  return launch_gemm(p, true, false, (hipStream_t)stream);
}

DVAE_API int dvae_conv5_wgrad(const float* dY, const float* X, float* dWp, int R, int N, int Cin, int Cout,
                              int split_k, void* stream) {
  GemmParams p{};
  p.A = dY; p.B = X; p.C = dWp; p.bias = nullptr;
  p.M = Cout; p.N = Cin; p.K = R;
  p.lda = Cout; p.ldb = Cin; p.ldc = Cin;
  p.taps = 5; p.tap_mode = 2;
  p.bk_row_shift = N; p.c_tap_stride = (int64_t)Cout * Cin;
  p.split_k = split_k; p.act = DVAE_ACT_NONE; p.epi = DVAE_EPI_ATOMIC;
  return launch_gemm(p, false, false, (hipStream_t)stream);
}
