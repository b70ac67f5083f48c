// Training-mode BatchNorm1d (+ReLU/tanh, + residual) on frame-major [R, C] data, forward and backward.
// HBM/L2-bound passes: 16-byte vector accesses along the channel axis, per-thread fp64 partial sums
// (the only place the path leaves fp32: it makes the one-pass E[y^2]-E[y]^2 variance exact enough
// to match a two-pass fp32 reference), wavefront-free cross-row reduction through LDS, deterministic
// two-stage reduction (partials -> finalize) instead of atomics.
//
// Group semantics: row r belongs to segment n = r % N and group g = n / (N/G).  The reference calls
// encode()/decode()/postnet() once per utterance of the pair, so each call has its own batch statistics
// and updates the running statistics once: G = 2 here reproduces that, in call order.
#include "common.h"

namespace {

constexpr int ROWS_PER_CHUNK = DVAE_BN_ROWS_PER_CHUNK;   // 512 workgroups at R = 16384, C = 512: two per CU (128 rows: one per CU at 2-3 TB/s; 32 rows: no faster, finalize slower)

__host__ __device__ inline int n_chunks(int R) { return (R + ROWS_PER_CHUNK - 1) / ROWS_PER_CHUNK; }

// partial[chunk][g][c][2] (fp64): sum and sum of squares (MODE 0), or sum(dU) and sum(dU*yhat) (MODE 1)
// 64 channel-quads x 4 row lanes per workgroup, 16-byte loads; fp64 accumulation per thread.
// ZB16: Z (the block's output, needed for the activation derivative) is stored as bf16 (bf16 compute mode)
// ZRE (MODE 1, ReLU / no activation): Z is not read — the activation's derivative is recomputed from Y with the forward
// pass's own expression (bn_apply_kernel), 8 instead of 12 bytes per element
template <int MODE, bool ZB16 = false, bool ZRE = false>
__global__ __launch_bounds__(256) void bn_partial_kernel(const float* __restrict__ Y, const float* __restrict__ dZ,
                                                         const void* __restrict__ Z, const float* __restrict__ mean,
                                                         const float* __restrict__ rstd, double* __restrict__ part,
                                                         int R, int N, int C, int G, int act,
                                                         const float* __restrict__ gamma = nullptr,
                                                         const float* __restrict__ beta = nullptr) {
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int c = (blockIdx.x * 64 + cl) * 4;       // C % 4 == 0
  const int r0 = blockIdx.y * ROWS_PER_CHUNK;
  const int r1 = min(R, r0 + ROWS_PER_CHUNK);
  const int per = N / G;
  double s[2][2][4];   // [group][stat][lane-of-quad]; indices are compile-time after unrolling
#pragma unroll
  for (int g = 0; g < 2; ++g)
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int k = 0; k < 4; ++k) s[g][q][k] = 0.0;
  if (c < C) {
    f32x4 mu0 = {0.f, 0.f, 0.f, 0.f}, mu1 = mu0, rs0 = {1.f, 1.f, 1.f, 1.f}, rs1 = rs0;
    if (MODE == 1) {
      mu0 = *reinterpret_cast<const f32x4*>(mean + c);
      rs0 = *reinterpret_cast<const f32x4*>(rstd + c);
      if (G > 1) {
        mu1 = *reinterpret_cast<const f32x4*>(mean + C + c);
        rs1 = *reinterpret_cast<const f32x4*>(rstd + C + c);
      }
    }
    f32x4 ga = rs0, be = mu0;
    if constexpr (ZRE) {
      ga = *reinterpret_cast<const f32x4*>(gamma + c);
      be = *reinterpret_cast<const f32x4*>(beta + c);
    }
    // (four rows' loads in flight per thread: with one row per trip the pass is bound by the load latency, not by HBM)
#pragma unroll 4
    for (int r = r0 + rl; r < r1; r += 4) {
      const bool g1 = ((r % N) / per) != 0;
      const f32x4 y = *reinterpret_cast<const f32x4*>(Y + (int64_t)r * C + c);
      f32x4 dz = y, z = y;
      if (MODE == 1) {
        dz = *reinterpret_cast<const f32x4*>(dZ + (int64_t)r * C + c);
        if constexpr (!ZRE) z = ld4<ZB16>(Z, ((int64_t)r * C + c) >> 2);
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        double v0, v1;
        if (MODE == 0) {
          v0 = (double)y[k];
          v1 = (double)y[k] * (double)y[k];
        } else {
          const float yh = (y[k] - (g1 ? mu1[k] : mu0[k])) * (g1 ? rs1[k] : rs0[k]);
          const float du = dz[k] * act_grad_from_out(ZRE ? act_apply(yh * ga[k] + be[k], act) : z[k], act);
          v0 = (double)du;
          v1 = (double)du * (double)yh;
        }
        if (g1) {
          s[1][0][k] += v0;
          s[1][1][k] += v1;
        } else {
          s[0][0][k] += v0;
          s[0][1][k] += v1;
        }
      }
    }
  }
  __shared__ double red[4][64][4];   // reused per (group, stat)
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    if (g >= G) break;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      __syncthreads();
#pragma unroll
      for (int k = 0; k < 4; ++k) red[rl][cl][k] = s[g][q][k];
      __syncthreads();
      if (rl == 0 && c < C) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const double tot = red[0][cl][k] + red[1][cl][k] + red[2][cl][k] + red[3][cl][k];
          part[(((int64_t)blockIdx.y * G + g) * C + c + k) * 2 + q] = tot;
        }
      }
    }
  }
}

// Finalize kernels: 4 channels per 256-thread workgroup (C/4 workgroups); thread (kl = tid >> 2, ci = tid & 3) sums the
// partials of channel ci over the chunks kl, kl + 64, ...: the four channels of a chunk are one 64-byte line, so a wave's
// load touches 16 lines (one wave per channel with its lanes striding over chunks touched 64, every line four times:
// 17 us per call at 1024 chunks x 512 channels).  Four chunks' loads are in flight per thread.  Shuffles over lane bits
// 2-5, then the four waves meet in LDS; the result is valid in threads 0..3.
__device__ __forceinline__ void sum_partials(const double* __restrict__ part, int chunks, int C, int G, int c, int kl,
                                             double (&o)[4]) {
  typedef double d2 __attribute__((ext_vector_type(2)));
  __shared__ double red[4][4][4];
  double s[4] = {0.0, 0.0, 0.0, 0.0};
  if (c < C) {
    const int64_t kstride = (int64_t)G * C * 2;
    const double* p0 = part + (int64_t)c * 2;
    int k = kl;
    for (; k + 192 < chunks; k += 256) {
      d2 v[4][2];
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int g = 0; g < 2; ++g)
          if (g < G) v[u][g] = *reinterpret_cast<const d2*>(p0 + (k + 64 * u) * kstride + (int64_t)g * C * 2);
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int g = 0; g < 2; ++g)
          if (g < G) {
            s[2 * g] += v[u][g][0];
            s[2 * g + 1] += v[u][g][1];
          }
    }
    for (; k < chunks; k += 64)
#pragma unroll
      for (int g = 0; g < 2; ++g)
        if (g < G) {
          const d2 v = *reinterpret_cast<const d2*>(p0 + k * kstride + (int64_t)g * C * 2);
          s[2 * g] += v[0];
          s[2 * g + 1] += v[1];
        }
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
#pragma unroll
    for (int off = 4; off < 64; off <<= 1) s[q] += __shfl_xor(s[q], off, 64);
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane < 4) {
#pragma unroll
    for (int q = 0; q < 4; ++q) red[wave][lane][q] = s[q];
  }
  __syncthreads();
  if (threadIdx.x < 4) {
#pragma unroll
    for (int q = 0; q < 4; ++q) o[q] = red[0][threadIdx.x][q] + red[1][threadIdx.x][q] + red[2][threadIdx.x][q] + red[3][threadIdx.x][q];
  }
}

__global__ __launch_bounds__(256) void bn_stats_finalize_kernel(const double* __restrict__ part,
                                                                float* __restrict__ mean, float* __restrict__ rstd,
                                                                float* __restrict__ rmean, float* __restrict__ rvar,
                                                                int64_t* __restrict__ nbt, int chunks, int R, int N,
                                                                int C, int G, float eps, float momentum) {
  const int kl = threadIdx.x >> 2, ci = threadIdx.x & 3;
  const int c = blockIdx.x * 4 + ci;
  double o[4];
  sum_partials(part, chunks, C, G, c, kl, o);
  if (threadIdx.x >= 4) return;
  if (c == 0 && nbt) *nbt += G;
  if (c >= C) return;
  const double cnt = (double)(R / N) * (double)(N / G);
  float rm = rmean ? rmean[c] : 0.f, rv = rvar ? rvar[c] : 0.f;
  for (int g = 0; g < G; ++g) {
    const double m = o[2 * g] / cnt;
    double var = o[2 * g + 1] / cnt - m * m;
    if (var < 0.0) var = 0.0;
    mean[g * C + c] = (float)m;
    rstd[g * C + c] = (float)(1.0 / sqrt(var + (double)eps));
    const float unb = (float)(cnt > 1.0 ? var * cnt / (cnt - 1.0) : var);
    rm = (1.f - momentum) * rm + momentum * (float)m;
    rv = (1.f - momentum) * rv + momentum * unb;
  }
  if (rmean) rmean[c] = rm;
  if (rvar) rvar[c] = rv;
}

template <bool ZB16>
__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ Y, const float* __restrict__ mean,
                                                       const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, const float* __restrict__ res,
                                                       void* __restrict__ Z, int64_t total4, int N, int C, int G,
                                                       int act) {
  const int c4n = C >> 2;
  const int per = N / G;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / c4n;
    const int c = (int)(i - r * c4n) * 4;
    const int g = (int)(r % N) / per;
    const f32x4 y = *reinterpret_cast<const f32x4*>(Y + i * 4);
    const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + g * C + c);
    const f32x4 rs = *reinterpret_cast<const f32x4*>(rstd + g * C + c);
    const f32x4 ga = *reinterpret_cast<const f32x4*>(gamma + c);
    const f32x4 be = *reinterpret_cast<const f32x4*>(beta + c);
    f32x4 z;
#pragma unroll
    for (int k = 0; k < 4; ++k) z[k] = act_apply((y[k] - mu[k]) * rs[k] * ga[k] + be[k], act);
    if (res) {
      const f32x4 rr = *reinterpret_cast<const f32x4*>(res + i * 4);
#pragma unroll
      for (int k = 0; k < 4; ++k) z[k] += rr[k];
    }
    st4<ZB16>(Z, i, z);
  }
}

// s12[g][c][2] = (sum dU, sum dU*yhat) as float; dgamma += sum_g s2 ; dbeta += sum_g s1
__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const double* __restrict__ part, float* __restrict__ s12,
                                                              float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                              int chunks, int C, int G) {
  const int kl = threadIdx.x >> 2, ci = threadIdx.x & 3;
  const int c = blockIdx.x * 4 + ci;
  double o[4];
  sum_partials(part, chunks, C, G, c, kl, o);
  if (threadIdx.x >= 4 || c >= C) return;
  double tg = 0.0, tb = 0.0;
  for (int g = 0; g < G; ++g) {
    s12[(g * C + c) * 2] = (float)o[2 * g];
    s12[(g * C + c) * 2 + 1] = (float)o[2 * g + 1];
    tb += o[2 * g];
    tg += o[2 * g + 1];
  }
  if (dgamma) dgamma[c] += (float)tg;
  if (dbeta) dbeta[c] += (float)tb;
}

template <bool ZB16, bool DYB16, bool ZRE = false>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* dZ, const float* __restrict__ Y,
                                                           const void* __restrict__ Z, const float* __restrict__ mean,
                                                           const float* __restrict__ rstd,
                                                           const float* __restrict__ gamma,
                                                           const float* __restrict__ s12, void* dY,
                                                           int64_t total4, int R, int N, int C, int G, int act,
                                                           const float* __restrict__ beta = nullptr) {
  const int c4n = C >> 2;
  const int per = N / G;
  const float inv_cnt = 1.f / ((float)(R / N) * (float)per);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / c4n;
    const int c = (int)(i - r * c4n) * 4;
    const int g = (int)(r % N) / per;
    const f32x4 dz = *reinterpret_cast<const f32x4*>(dZ + i * 4);
    const f32x4 y = *reinterpret_cast<const f32x4*>(Y + i * 4);
    f32x4 z = y, be = y;
    if constexpr (ZRE) be = *reinterpret_cast<const f32x4*>(beta + c);
    else z = ld4<ZB16>(Z, i);
    const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + g * C + c);
    const f32x4 rs = *reinterpret_cast<const f32x4*>(rstd + g * C + c);
    const f32x4 ga = *reinterpret_cast<const f32x4*>(gamma + c);
    f32x4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float yh = (y[k] - mu[k]) * rs[k];
      const float du = dz[k] * act_grad_from_out(ZRE ? act_apply(yh * ga[k] + be[k], act) : z[k], act);
      const float s1 = s12[(g * C + c + k) * 2], s2 = s12[(g * C + c + k) * 2 + 1];
      o[k] = ga[k] * rs[k] * (du - s1 * inv_cnt - yh * s2 * inv_cnt);
    }
    st4<DYB16>(dY, i, o);
  }
}

int check(int R, int N, int C, int G) {
  if (R <= 0 || N <= 0 || C <= 0 || G < 1 || G > 2) return DVAE_EINVAL;
  if ((R % N) || (N % G) || (C & 3)) return DVAE_EINVAL;
  return DVAE_OK;
}

}  // namespace

DVAE_API int64_t dvae_bn_ws_bytes(int R, int C, int G) {
  // partials + s12
  return (int64_t)n_chunks(R) * G * C * 2 * sizeof(double) + (int64_t)G * C * 2 * sizeof(float) + 64;
}

DVAE_API int dvae_bn_stats_fwd(const float* Y, float* mean, float* rstd, float* running_mean, float* running_var,
                               int64_t* num_batches_tracked, void* ws, int R, int N, int C, int G, float eps,
                               float momentum, void* stream) {
  if (check(R, N, C, G) || !Y || !mean || !rstd || !ws) return DVAE_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  const int ch = n_chunks(R);
  double* part = (double*)ws;
  dim3 grid((C + 255) / 256, ch);
  hipLaunchKernelGGL((bn_partial_kernel<0, false>), grid, dim3(256), 0, s, Y, nullptr, nullptr, nullptr, nullptr, part, R, N,
                     C, G, 0);
  hipLaunchKernelGGL(bn_stats_finalize_kernel, dim3((C + 3) / 4), dim3(256), 0, s, part, mean, rstd,
                     running_mean, running_var, num_batches_tracked, ch, R, N, C, G, eps, momentum);
  return dvae_check_launch();
}

// the second half of dvae_bn_stats_fwd: the partial sums are already in `ws` (dvae_conv5_fwd_stats)
DVAE_API int dvae_bn_stats_finalize(float* mean, float* rstd, float* running_mean, float* running_var,
                                    int64_t* num_batches_tracked, const void* ws, int R, int N, int C, int G, float eps,
                                    float momentum, void* stream) {
  if (check(R, N, C, G) || !mean || !rstd || !ws) return DVAE_EINVAL;
  hipLaunchKernelGGL(bn_stats_finalize_kernel, dim3((C + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const double*)ws,
                     mean, rstd, running_mean, running_var, num_batches_tracked, n_chunks(R), R, N, C, G, eps, momentum);
  return dvae_check_launch();
}

DVAE_API int dvae_bn_apply_fwd(const float* Y, const float* mean, const float* rstd, const float* gamma,
                               const float* beta, const float* residual, void* Z, int R, int N, int C, int G, int act,
                               int z_bf16, void* stream) {
  if (check(R, N, C, G) || !Y || !mean || !rstd || !gamma || !beta || !Z) return DVAE_EINVAL;
  const int64_t total4 = (int64_t)R * C / 4;
  const int blocks = (int)((total4 + 255) / 256 < 2048 ? (total4 + 255) / 256 : 2048);
  if (z_bf16)
    hipLaunchKernelGGL(bn_apply_kernel<true>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, Y, mean, rstd, gamma, beta,
                       residual, Z, total4, N, C, G, act);
  else
    hipLaunchKernelGGL(bn_apply_kernel<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, Y, mean, rstd, gamma, beta,
                       residual, Z, total4, N, C, G, act);
  return dvae_check_launch();
}

namespace {
int bn_bwd_launch(const float* dZ, const float* Y, const void* Z, const float* mean, const float* rstd,
                  const float* gamma, const float* beta, void* dY, float* dgamma, float* dbeta, void* ws, int R, int N,
                  int C, int G, int act, int dtypes, void* stream) {
  const bool zb = dtypes & 1, dyb = dtypes & 2, zre = (Z == nullptr);
  if (check(R, N, C, G) || !dZ || !Y || !mean || !rstd || !gamma || !dY || !ws) return DVAE_EINVAL;
  // without Z the activation's derivative is recomputed from Y: ReLU (a compare) or none — not tanh (a libm call per
  // element costs these HBM-bound passes more than the 4 bytes it saves)
  if (zre && (!beta || (act != DVAE_ACT_RELU && act != DVAE_ACT_NONE))) return DVAE_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  const int ch = n_chunks(R);
  double* part = (double*)ws;
  float* s12 = (float*)((char*)ws + (int64_t)ch * G * C * 2 * sizeof(double));
  dim3 grid((C + 255) / 256, ch);
  if (zre) hipLaunchKernelGGL((bn_partial_kernel<1, false, true>), grid, dim3(256), 0, s, Y, dZ, Z, mean, rstd, part, R, N, C, G, act, gamma, beta);
  else if (zb) hipLaunchKernelGGL((bn_partial_kernel<1, true>), grid, dim3(256), 0, s, Y, dZ, Z, mean, rstd, part, R, N, C, G, act);
  else hipLaunchKernelGGL((bn_partial_kernel<1, false>), grid, dim3(256), 0, s, Y, dZ, Z, mean, rstd, part, R, N, C, G, act);
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + 3) / 4), dim3(256), 0, s, part, s12, dgamma, dbeta, ch, C,
                     G);
  const int64_t total4 = (int64_t)R * C / 4;
  const int blocks = (int)((total4 + 255) / 256 < 2048 ? (total4 + 255) / 256 : 2048);
#define BWDA(ZB_, DYB_, ZRE_) hipLaunchKernelGGL((bn_bwd_apply_kernel<ZB_, DYB_, ZRE_>), dim3(blocks), dim3(256), 0, s, dZ, Y, Z, mean, rstd, gamma, s12, dY, total4, R, N, C, G, act, beta)
  if (zre) { if (dyb) BWDA(false, true, true); else BWDA(false, false, true); }
  else if (zb && dyb) BWDA(true, true, false); else if (zb) BWDA(true, false, false);
  else if (dyb) BWDA(false, true, false); else BWDA(false, false, false);
#undef BWDA
  return dvae_check_launch();
}
}  // namespace

DVAE_API int dvae_bn_bwd(const float* dZ, const float* Y, const void* Z, const float* mean, const float* rstd,
                         const float* gamma, void* dY, float* dgamma, float* dbeta, void* ws, int R, int N, int C,
                         int G, int act, int dtypes, void* stream) {
  if (!Z) return DVAE_EINVAL;
  return bn_bwd_launch(dZ, Y, Z, mean, rstd, gamma, nullptr, dY, dgamma, dbeta, ws, R, N, C, G, act, dtypes, stream);
}
DVAE_API int dvae_bn_bwd_from_y(const float* dZ, const float* Y, const float* mean, const float* rstd,
                                const float* gamma, const float* beta, void* dY, float* dgamma, float* dbeta, void* ws,
                                int R, int N, int C, int G, int act, int dtypes, void* stream) {
  if (!beta) return DVAE_EINVAL;
  return bn_bwd_launch(dZ, Y, nullptr, mean, rstd, gamma, beta, dY, dgamma, dbeta, ws, R, N, C, G, act, dtypes, stream);
}
