// The HBM-bound and tiny kernels of the step: reparameterise + latent assembly, KL and L1 reductions
// (wavefront shuffle reductions, fp64 only for the final cross-workgroup sums), flat Adam, layout changes.
#include <type_traits>
#include "common.h"

int g_dvae_last_hip_error = 0;

DVAE_API int dvae_version(void) { return DVAE_ABI_VERSION; }
DVAE_API int dvae_last_hip_error(void) { return g_dvae_last_hip_error; }

namespace {

// ------------------------------------------------------------------ latent
__global__ void latent_fwd_kernel(const float* __restrict__ style, const float* __restrict__ content,
                                  const float* __restrict__ eps_c, const float* __restrict__ eps_s,
                                  float* __restrict__ z, float* __restrict__ q_mu, float* __restrict__ q_lv,
                                  float* __restrict__ s_mu, float* __restrict__ s_lv, int Bh, int S, int Cn) {
  const int D = S + Cn;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= 2 * Bh * D) return;
  const int n = idx / D, dcol = idx - n * D;
  const int b = n % Bh;
  if (dcol < S) {
    const float mu = 0.5f * (style[b * 2 * S + dcol] + style[(Bh + b) * 2 * S + dcol]);
    const float lv = 0.5f * (style[b * 2 * S + S + dcol] + style[(Bh + b) * 2 * S + S + dcol]);
    q_mu[idx] = mu;
    q_lv[idx] = lv;
    z[idx] = eps_s[b * S + dcol] * expf(0.5f * lv) + mu;
    if (n < Bh) {
      s_mu[b * S + dcol] = mu;
      s_lv[b * S + dcol] = lv;
    }
  } else {
    const int k = dcol - S;
    const float mu = content[n * 2 * Cn + k];
    const float lv = content[n * 2 * Cn + Cn + k];
    q_mu[idx] = mu;
    q_lv[idx] = lv;
    z[idx] = eps_c ? eps_c[n * Cn + k] * expf(0.5f * lv) + mu : mu;
  }
}

__global__ void latent_bwd_kernel(const float* __restrict__ style, const float* __restrict__ content,
                                  const float* __restrict__ eps_c, const float* __restrict__ eps_s,
                                  const float* __restrict__ dz, const float* __restrict__ dq_mu,
                                  const float* __restrict__ dq_lv, const float* __restrict__ ds_mu,
                                  const float* __restrict__ ds_lv, float* __restrict__ dstyle,
                                  float* __restrict__ dcontent, int Bh, int S, int Cn) {
  const int D = S + Cn;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= 2 * Bh * D) return;
  const int n = idx / D, dcol = idx - n * D;
  const int b = n % Bh;
  auto ld = [](const float* p, int i) { return p ? p[i] : 0.f; };
  if (dcol < S) {
    if (n >= Bh) {  // x2's style head is detached (disentangled_vae.py:257-258)
      dstyle[n * 2 * S + dcol] = 0.f;
      dstyle[n * 2 * S + S + dcol] = 0.f;
      return;
    }
    const int i1 = b * D + dcol, i2 = (Bh + b) * D + dcol;
    const float lv = 0.5f * (style[b * 2 * S + S + dcol] + style[(Bh + b) * 2 * S + S + dcol]);
    const float gz = ld(dz, i1) + ld(dz, i2);
    const float gmu = gz + ld(dq_mu, i1) + ld(dq_mu, i2) + ld(ds_mu, b * S + dcol);
    const float glv = gz * eps_s[b * S + dcol] * 0.5f * expf(0.5f * lv) + ld(dq_lv, i1) + ld(dq_lv, i2) +
                      ld(ds_lv, b * S + dcol);
    dstyle[b * 2 * S + dcol] = 0.5f * gmu;
    dstyle[b * 2 * S + S + dcol] = 0.5f * glv;
  } else {
    const int k = dcol - S;
    const float lv = content[n * 2 * Cn + Cn + k];
    const float gz = ld(dz, idx);
    dcontent[n * 2 * Cn + k] = gz + ld(dq_mu, idx);
    dcontent[n * 2 * Cn + Cn + k] =
        (eps_c ? gz * eps_c[n * Cn + k] * 0.5f * expf(0.5f * lv) : 0.f) + ld(dq_lv, idx);
  }
}

// ------------------------------------------------------------------ KL
__global__ __launch_bounds__(256) void kl_fwd_kernel(const float* __restrict__ mu, const float* __restrict__ lv,
                                                     float* __restrict__ out, int64_t n, float scale) {
  double acc = 0.0;
  for (int64_t i = threadIdx.x; i < n; i += 256) {
    const float m = mu[i], l = lv[i];
    acc += (double)(1.f + l - m * m - expf(l));
  }
  acc = wave_sum_d(acc);
  __shared__ double red[4];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = (float)((red[0] + red[1] + red[2] + red[3]) * (double)scale);
}

__global__ void kl_bwd_kernel(const float* __restrict__ mu, const float* __restrict__ lv,
                              const float* __restrict__ gout, float* __restrict__ dmu, float* __restrict__ dlv,
                              int64_t n, float scale) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float g = gout[0] * scale;
  dmu[i] = g * (-2.f * mu[i]);
  dlv[i] = g * (1.f - expf(lv[i]));
}

// ------------------------------------------------------------------ L1
constexpr int L1_BLOCKS = 512;

__global__ __launch_bounds__(256) void l1_partial_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                         double* __restrict__ part, int64_t n) {
  const int64_t n4 = n >> 2;
  float acc = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(x + 4 * i);
    const f32x4 b = *reinterpret_cast<const f32x4*>(y + 4 * i);
    acc += fabsf(a[0] - b[0]) + fabsf(a[1] - b[1]) + fabsf(a[2] - b[2]) + fabsf(a[3] - b[3]);
  }
  if (blockIdx.x == 0)
    for (int64_t i = (n4 << 2) + threadIdx.x; i < n; i += 256) acc += fabsf(x[i] - y[i]);
  double d = wave_sum_d((double)acc);
  __shared__ double red[4];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = d;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(64) void l1_final_kernel(const double* __restrict__ part, float* __restrict__ out,
                                                      int nparts, float scale) {
  double a = 0.0;
  for (int i = threadIdx.x; i < nparts; i += 64) a += part[i];
  a = wave_sum_d(a);
  if (threadIdx.x == 0) out[0] = (float)(a * (double)scale);
}

__global__ __launch_bounds__(256) void l1_bwd_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                     const float* __restrict__ gout, float* __restrict__ dy,
                                                     int64_t n, float scale) {
  const float g = gout[0] * scale;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float d = x[i] - y[i];
    dy[i] = d > 0.f ? -g : (d < 0.f ? g : 0.f);
  }
}

// ------------------------------------------------------------------ the whole loss_functionGVAE2 in two launches
// (disentangled_vae.py:310-327): four L1 sums / batch_size, two KL terms, the report-only style KL and the weighted
// total.  Round 1 spent 19 launches on it forward (4 x (partial + final), 3 KL, ~8 ATen scalar ops) and ~15 backward.
struct LossArgs {
  const float* x[2];        // x1, x2                      [n]
  const float* r[4];        // recon1, recon2, recon1_hat, recon2_hat   [n]
  const float* qmu[2];      // q_z1_mu, q_z2_mu            [nq]
  const float* qlv[2];
  const float* smu;         // z_style_mu, z_style_logvar  [ns]
  const float* slv;
  int64_t n;
  int nq, ns;
  float l1_scale, kl_scale, style_scale, mse_cof, kl_cof;
};

__global__ __launch_bounds__(256) void loss_partial_kernel(const LossArgs a, double* __restrict__ part) {
  const int64_t n4 = a.n >> 2;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const f32x4 x1 = *reinterpret_cast<const f32x4*>(a.x[0] + 4 * i), x2 = *reinterpret_cast<const f32x4*>(a.x[1] + 4 * i);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const f32x4 r = *reinterpret_cast<const f32x4*>(a.r[k] + 4 * i);
      const f32x4 x = (k & 1) ? x2 : x1;
      acc[k] += fabsf(x[0] - r[0]) + fabsf(x[1] - r[1]) + fabsf(x[2] - r[2]) + fabsf(x[3] - r[3]);
    }
  }
  if (blockIdx.x == 0)
    for (int64_t i = (n4 << 2) + threadIdx.x; i < a.n; i += 256)
#pragma unroll
      for (int k = 0; k < 4; ++k) acc[k] += fabsf(a.x[k & 1][i] - a.r[k][i]);
  __shared__ double red[4][4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const double d = wave_sum_d((double)acc[k]);
    if ((threadIdx.x & 63) == 0) red[k][threadIdx.x >> 6] = d;
  }
  __syncthreads();
  if (threadIdx.x < 4)
    part[blockIdx.x * 4 + threadIdx.x] = red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3];
}

// out[8] = (LOSS, L1_x1, L1_x2, L1_x1hat, L1_x2hat, KL_z1, KL_z2, KL_style)
__global__ __launch_bounds__(256) void loss_final_kernel(const LossArgs a, const double* __restrict__ part, int nparts,
                                                         float* __restrict__ out) {
  __shared__ double red[7][4];
  double v[7] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  for (int i = threadIdx.x; i < nparts; i += 256)
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] += part[i * 4 + k];
  auto kl = [](float m, float l) { return (double)(1.f + l - m * m - expf(l)); };
  for (int i = threadIdx.x; i < a.nq; i += 256) {
    v[4] += kl(a.qmu[0][i], a.qlv[0][i]);
    v[5] += kl(a.qmu[1][i], a.qlv[1][i]);
  }
  for (int i = threadIdx.x; i < a.ns; i += 256) v[6] += kl(a.smu[i], a.slv[i]);
#pragma unroll
  for (int k = 0; k < 7; ++k) {
    const double d = wave_sum_d(v[k]);
    if ((threadIdx.x & 63) == 0) red[k][threadIdx.x >> 6] = d;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float t[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) {
      const double tot = red[k][0] + red[k][1] + red[k][2] + red[k][3];
      t[k] = (float)(tot * (double)(k < 4 ? a.l1_scale : (k < 6 ? a.kl_scale : a.style_scale)));
      out[1 + k] = t[k];
    }
    // the reference's operation order: mse_cof * (((a + b) + c) + d) + kl_cof * (e + f), all in fp32
    out[0] = a.mse_cof * (((t[0] + t[1]) + t[2]) + t[3]) + a.kl_cof * (t[4] + t[5]);
  }
}

// gradients w.r.t. the four reconstructions and the six latent statistics, given g[8] = dL/d(out[8])
__global__ __launch_bounds__(256) void loss_bwd_kernel(const LossArgs a, const float* __restrict__ g, float* __restrict__ dr0,
                                                       float* __restrict__ dr1, float* __restrict__ dr2,
                                                       float* __restrict__ dr3, float* __restrict__ dqmu0,
                                                       float* __restrict__ dqlv0, float* __restrict__ dqmu1,
                                                       float* __restrict__ dqlv1, float* __restrict__ dsmu,
                                                       float* __restrict__ dslv, int l1_blocks) {
  const float g0 = g[0];
  if ((int)blockIdx.x < l1_blocks) {
    float* dr[4] = {dr0, dr1, dr2, dr3};
    float w[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) w[k] = (g0 * a.mse_cof + g[1 + k]) * a.l1_scale;
    const int64_t n4 = a.n >> 2;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)l1_blocks * 256) {
      const f32x4 x1 = *reinterpret_cast<const f32x4*>(a.x[0] + 4 * i), x2 = *reinterpret_cast<const f32x4*>(a.x[1] + 4 * i);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (!dr[k]) continue;
        const f32x4 r = *reinterpret_cast<const f32x4*>(a.r[k] + 4 * i);
        const f32x4 x = (k & 1) ? x2 : x1;
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float d = x[e] - r[e];
          o[e] = d > 0.f ? -w[k] : (d < 0.f ? w[k] : 0.f);
        }
        *reinterpret_cast<f32x4*>(dr[k] + 4 * i) = o;
      }
    }
    if (blockIdx.x == 0)
      for (int64_t i = (n4 << 2) + threadIdx.x; i < a.n; i += 256)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          if (!dr[k]) continue;
          const float d = a.x[k & 1][i] - a.r[k][i];
          dr[k][i] = d > 0.f ? -w[k] : (d < 0.f ? w[k] : 0.f);
        }
    return;
  }
  // latent statistics: one extra workgroup
  const float w1 = (g0 * a.kl_cof + g[5]) * a.kl_scale, w2 = (g0 * a.kl_cof + g[6]) * a.kl_scale, ws = g[7] * a.style_scale;
  for (int i = threadIdx.x; i < a.nq; i += 256) {
    if (dqmu0) { dqmu0[i] = w1 * (-2.f * a.qmu[0][i]); dqlv0[i] = w1 * (1.f - expf(a.qlv[0][i])); }
    if (dqmu1) { dqmu1[i] = w2 * (-2.f * a.qmu[1][i]); dqlv1[i] = w2 * (1.f - expf(a.qlv[1][i])); }
  }
  if (dsmu)
    for (int i = threadIdx.x; i < a.ns; i += 256) {
      dsmu[i] = ws * (-2.f * a.smu[i]);
      dslv[i] = ws * (1.f - expf(a.slv[i]));
    }
}

// ------------------------------------------------------------------ Adam
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, int64_t n, float lr,
                                                   float b1, float b2, float eps, float gs, float bc1, float bc2s) {
  // torch.optim.Adam (no amsgrad, no weight decay):
  //   m += (g-m)(1-b1) ; v = b2 v + (1-b2) g^2 ; p -= (lr/bc1) * m / (sqrt(v)/sqrt(bc2) + eps)
  const int64_t n4 = n >> 2;
  const float step_size = lr / bc1;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    f32x4 pp = *reinterpret_cast<f32x4*>(p + 4 * i);
    const f32x4 gg = *reinterpret_cast<const f32x4*>(g + 4 * i);
    f32x4 mm = *reinterpret_cast<f32x4*>(m + 4 * i);
    f32x4 vv = *reinterpret_cast<f32x4*>(v + 4 * i);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float gr = gg[k] * gs;
      mm[k] = mm[k] + (gr - mm[k]) * (1.f - b1);
      vv[k] = vv[k] * b2 + (1.f - b2) * gr * gr;
      pp[k] -= step_size * (mm[k] / (sqrtf(vv[k]) / bc2s + eps));
    }
    *reinterpret_cast<f32x4*>(p + 4 * i) = pp;
    *reinterpret_cast<f32x4*>(m + 4 * i) = mm;
    *reinterpret_cast<f32x4*>(v + 4 * i) = vv;
  }
  if (blockIdx.x == 0) {
    for (int64_t i = (n4 << 2) + threadIdx.x; i < n; i += 256) {
      const float gr = g[i] * gs;
      const float mk = m[i] + (gr - m[i]) * (1.f - b1);
      const float vk = v[i] * b2 + (1.f - b2) * gr * gr;
      m[i] = mk;
      v[i] = vk;
      p[i] -= step_size * (mk / (sqrtf(vk) / bc2s + eps));
    }
  }
}

// Device-resident step counter, bias corrections, learning rate and gradient scale, so that a captured hipGraph replays a
// correct Adam step: state[0] = t (as float), state[1] = 1 - beta1^t, state[2] = sqrt(1 - beta2^t), state[4] = lr,
// state[5] = grad_scale.  `skip`: while *skip != 0 (sticky error word of the persistent LSTM launches) nothing is touched.
__global__ void adam_tick_kernel(float* __restrict__ state, float b1, float b2, const unsigned* __restrict__ skip) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    if (skip && __hip_atomic_load(skip, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
    const double t = (double)state[0] + 1.0;
    state[0] = (float)t;
    state[1] = (float)(1.0 - pow((double)b1, t));
    state[2] = (float)sqrt(1.0 - pow((double)b2, t));
  }
}
struct AdamClear {
  int64_t lo[8], hi[8];
  int n;
};
template <int U>
__global__ __launch_bounds__(256) void adam_dev_kernel(float* __restrict__ p, float* __restrict__ g,
                                                       float* __restrict__ m, float* __restrict__ v, int64_t n4,
                                                       float b1, float b2, float eps, const float* __restrict__ state,
                                                       const unsigned* __restrict__ skip, AdamClear clr) {
  if (skip && __hip_atomic_load(skip, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
  const float bc1 = state[1], bc2s = state[2], lr = state[4], gs = state[5];
  const float step_size = lr / bc1;
  const f32x4 z = {0.f, 0.f, 0.f, 0.f};
  // U independent 16-byte accesses per tensor and thread per trip: 4 U loads in flight before the first use
  for (int64_t i0 = (int64_t)blockIdx.x * (256 * U) + threadIdx.x; i0 < n4; i0 += (int64_t)gridDim.x * (256 * U)) {
    f32x4 pp[U], gg[U], mm[U], vv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t i = i0 + u * 256 < n4 ? i0 + u * 256 : i0;      // clamped: the tail re-reads element i0, stores are guarded
      // streamed once per step: non-temporal accesses keep 2.7 GB of optimiser traffic from evicting the caches
      pp[u] = __builtin_nontemporal_load(reinterpret_cast<f32x4*>(p + 4 * i));
      gg[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(g + 4 * i));
      mm[u] = __builtin_nontemporal_load(reinterpret_cast<f32x4*>(m + 4 * i));
      vv[u] = __builtin_nontemporal_load(reinterpret_cast<f32x4*>(v + 4 * i));
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t i = i0 + u * 256;
      if (i >= n4) break;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float gr = gg[u][k] * gs;
        mm[u][k] = mm[u][k] + (gr - mm[u][k]) * (1.f - b1);
        vv[u][k] = vv[u][k] * b2 + (1.f - b2) * gr * gr;
        pp[u][k] -= step_size * (mm[u][k] / (sqrtf(vv[u][k]) / bc2s + eps));
      }
      __builtin_nontemporal_store(pp[u], reinterpret_cast<f32x4*>(p + 4 * i));
      __builtin_nontemporal_store(mm[u], reinterpret_cast<f32x4*>(m + 4 * i));
      __builtin_nontemporal_store(vv[u], reinterpret_cast<f32x4*>(v + 4 * i));
      bool c = false;
      for (int r = 0; r < clr.n; ++r) c |= (4 * i >= clr.lo[r]) & (4 * i < clr.hi[r]);
      if (c) *reinterpret_cast<f32x4*>(g + 4 * i) = z;     // plain store: the next step's launches accumulate into it
    }
  }
}

// ------------------------------------------------------------------ layout
// X[t][g*Bh+b][c] = x_g[b][c][t] ; tile-transpose through LDS over (c,t) per segment
template <bool OB16>
__global__ __launch_bounds__(256) void mel_to_frames_kernel(const float* __restrict__ x1,
                                                            const float* __restrict__ x2, void* __restrict__ Xv,
                                                            int Bh, int C, int T, int N) {
  using out_t = typename std::conditional<OB16, __bf16, float>::type;
  out_t* __restrict__ X = reinterpret_cast<out_t*>(Xv);
  __shared__ float tile[32][33];
  const int n = blockIdx.z;
  const float* __restrict__ src = (n < Bh ? x1 + (int64_t)n * C * T : x2 + (int64_t)(n - Bh) * C * T);
  const int c0 = blockIdx.y * 32, t0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int k = ty; k < 32; k += 8) {
    const int c = c0 + k, t = t0 + tx;
    tile[k][tx] = (c < C && t < T) ? src[(int64_t)c * T + t] : 0.f;
  }
  __syncthreads();
  for (int k = ty; k < 32; k += 8) {
    const int t = t0 + k, c = c0 + tx;
    if (t < T && c < C) X[((int64_t)t * N + n) * C + c] = (out_t)tile[tx][k];
  }
}

// out[n][c][t] = X[t][n][c]
__global__ __launch_bounds__(256) void frames_to_mel_kernel(const float* __restrict__ X, float* __restrict__ out,
                                                            int N, int C, int T) {
  __shared__ float tile[32][33];
  const int n = blockIdx.z;
  const int c0 = blockIdx.y * 32, t0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int k = ty; k < 32; k += 8) {
    const int t = t0 + k, c = c0 + tx;
    tile[k][tx] = (t < T && c < C) ? X[((int64_t)t * N + n) * C + c] : 0.f;
  }
  __syncthreads();
  for (int k = ty; k < 32; k += 8) {
    const int c = c0 + k, t = t0 + tx;
    if (c < C && t < T) out[((int64_t)n * C + c) * T + t] = tile[tx][k];
  }
}

__global__ __launch_bounds__(256) void permute_102_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                          int A, int B, int C4) {
  // in[a][b][c] -> out[b][a][c], float4 along c
  const int64_t total = (int64_t)A * B * C4;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % C4);
    const int64_t ab = i / C4;
    const int a = (int)(ab % A);
    const int b = (int)(ab / A);
    reinterpret_cast<f32x4*>(out)[i] = reinterpret_cast<const f32x4*>(in)[((int64_t)a * B + b) * C4 + c];
  }
}

// out[c] += sum_r X[r][c]: 64 column-quads x 4 row lanes per workgroup, 16-byte loads (1 KiB per wave per row)
template <bool XB16>
__global__ __launch_bounds__(256) void colsum_kernel(const void* __restrict__ Xv, float* __restrict__ o1,
                                                     float* __restrict__ o2, int R, int C, int64_t ld, int rows_pb) {
  using elem_t = typename std::conditional<XB16, __bf16, float>::type;
  const elem_t* __restrict__ X = reinterpret_cast<const elem_t*>(Xv);
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int c = (blockIdx.x * 64 + cl) * 4;
  const int r0 = blockIdx.y * rows_pb, r1 = min(R, r0 + rows_pb);
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  if (c + 3 < C) {
    // eight independent 16-byte loads in flight per thread (one dependent load per iteration left the kernel at
    // ~2 TB/s: too little memory-level parallelism for HBM latency)
    const elem_t* __restrict__ px = X + (int64_t)(r0 + rl) * ld + c;
    const int n = (r1 - r0 - rl + 3) >> 2;          // rows of this thread
    int i = 0;
    for (; i + 8 <= n; i += 8) {
      f32x4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = ld4<XB16>(px + (int64_t)(i + u) * 4 * ld, 0);
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; i < n; ++i) s += ld4<XB16>(px + (int64_t)i * 4 * ld, 0);
  } else if (c < C) {
    for (int r = r0 + rl; r < r1; r += 4)
      for (int k = 0; k < 4 && c + k < C; ++k) s[k] += (float)X[(int64_t)r * ld + c + k];
  }
  __shared__ f32x4 red[4][64];
  red[rl][cl] = s;
  __syncthreads();
  if (rl == 0 && c < C) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (c + k >= C) break;
      const float tot = red[0][cl][k] + red[1][cl][k] + red[2][cl][k] + red[3][cl][k];
      atomicAdd(o1 + c + k, tot);
      if (o2) atomicAdd(o2 + c + k, tot);
    }
  }
}

__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ in, float* __restrict__ out, int R,
                                                        int C) {
  __shared__ float tile[32][33];
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int k = ty; k < 32; k += 8) {
    const int r = r0 + k, c = c0 + tx;
    tile[k][tx] = (r < R && c < C) ? in[(int64_t)r * C + c] : 0.f;
  }
  __syncthreads();
  for (int k = ty; k < 32; k += 8) {
    const int c = c0 + k, r = r0 + tx;
    if (c < C && r < R) out[(int64_t)c * R + r] = tile[tx][k];
  }
}

// dU = dZ * act'(Z)   (Linear + ReLU backward, disentangled_vae.py:211)
__global__ __launch_bounds__(256) void act_bwd_kernel(const float* dz, const float* __restrict__ z, float* du,
                                                      int64_t n, int act) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
    du[i] = dz[i] * act_grad_from_out(z[i], act);
}

__global__ __launch_bounds__(256) void act_fwd_kernel(float* y, int64_t n, int act) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
    y[i] = act_apply(y[i], act);
}

// W[co][ci][5] -> Wp[tap][co][ci]
__global__ __launch_bounds__(256) void conv_pack_kernel(const float* __restrict__ W, float* __restrict__ Wp,
                                                        int64_t cc) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < cc; i += (int64_t)gridDim.x * 256) {
#pragma unroll
    for (int tap = 0; tap < 5; ++tap) Wp[tap * cc + i] = W[i * 5 + tap];
  }
}
// W[co][ci][5] -> Wpt[tap][ci][co]: the transposed pack the data-gradient contraction reads k-contiguously
// (32x32 tiles of the (co, ci) plane through LDS: coalesced on both sides up to the stride-5 gather)
__global__ __launch_bounds__(256) void conv_pack_t_kernel(const float* __restrict__ W, float* __restrict__ Wpt, int Cout,
                                                          int Cin) {
  __shared__ float tile[5][32][33];
  const int co0 = blockIdx.y * 32, ci0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int r = ty; r < 32; r += 8) {
    const int co = co0 + r, ci = ci0 + tx;
    if (co < Cout && ci < Cin) {
      const float* src = W + ((int64_t)co * Cin + ci) * 5;
#pragma unroll
      for (int tap = 0; tap < 5; ++tap) tile[tap][r][tx] = src[tap];
    }
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int ci = ci0 + r, co = co0 + tx;
    if (ci < Cin && co < Cout) {
#pragma unroll
      for (int tap = 0; tap < 5; ++tap) Wpt[((int64_t)tap * Cin + ci) * Cout + co] = tile[tap][tx][r];
    }
  }
}
__global__ __launch_bounds__(256) void conv_unpack_add_kernel(const float* __restrict__ dWp, float* __restrict__ dW,
                                                              int64_t cc) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < cc; i += (int64_t)gridDim.x * 256) {
#pragma unroll
    for (int tap = 0; tap < 5; ++tap) dW[i * 5 + tap] += dWp[tap * cc + i];
  }
}

// ---- inference-side layout helpers (mel -> mel conversion, variational_base_vae.py:269-298, 335-348)
__global__ __launch_bounds__(256) void mel_to_chunks_kernel(const float* __restrict__ mel, float* __restrict__ out,
                                                            int C, int L, int T, int n) {
  const int64_t total = (int64_t)n * C * T;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int t = (int)(i % T);
    const int c = (int)((i / T) % C);
    const int k = (int)(i / ((int64_t)T * C));
    const int64_t src = (int64_t)k * T + t;
    out[i] = src < L ? mel[(int64_t)c * L + src] : 0.f;
  }
}
__global__ __launch_bounds__(256) void chunks_to_mel_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                            int n, int C, int T, float lo, float hi, int clamp) {
  const int64_t W = (int64_t)n * T, total = (int64_t)C * W;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i / W);
    const int64_t col = i % W;
    float v = in[((col / T) * C + c) * T + col % T];
    if (clamp) v = fminf(fmaxf(v, lo), hi);
    out[i] = v;
  }
}
// z_src[k] = [mean_rows(src_style_mu) | src_content_mu[k]] ; z_conv[k] = [mean_rows(trg_style_mu) | src_content_mu[k]]
__global__ void conversion_latents_kernel(const float* __restrict__ ss, const float* __restrict__ sc,
                                          const float* __restrict__ ts, float* __restrict__ zs, float* __restrict__ zc,
                                          int n, int m, int S, int Cn) {
  const int D = S + Cn;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n * D) return;
  const int k = idx / D, d = idx % D;
  if (d < S) {
    float a = 0.f, b = 0.f;
    for (int r = 0; r < n; ++r) a += ss[r * 2 * S + d];
    for (int r = 0; r < m; ++r) b += ts[r * 2 * S + d];
    zs[idx] = a / (float)n;
    zc[idx] = b / (float)m;
  } else {
    const float v = sc[k * 2 * Cn + (d - S)];
    zs[idx] = v;
    zc[idx] = v;
  }
}
__global__ __launch_bounds__(256) void mul_div_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                      const float* __restrict__ c, float* __restrict__ out, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
    out[i] = a[i] * (b[i] / c[i]);
}

// out[i][c][t] = mels[utt[i]][c][off[i] + t] if off[i] + t < len[utt[i]] else 0   (random crop / right zero pad of
// preprocessing/dataset.py:100-109 for a whole batch, from a device-resident padded store [n_utt, C, Lmax])
__global__ __launch_bounds__(256) void gather_crop_kernel(const float* __restrict__ mels, const int* __restrict__ lens,
                                                          const int* __restrict__ utt, const int* __restrict__ off,
                                                          float* __restrict__ out, int n, int C, int T, int Lmax) {
  const int64_t total = (int64_t)n * C * T;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int t = (int)(i % T);
    const int c = (int)((i / T) % C);
    const int k = (int)(i / ((int64_t)T * C));
    const int u = utt[k];
    const int src = off[k] + t;
    out[i] = (src < lens[u]) ? mels[((int64_t)u * C + c) * Lmax + src] : 0.f;
  }
}

inline int nblk(int64_t n, int per = 256, int cap = 2048) {
  int64_t b = (n + per - 1) / per;
  if (b < 1) b = 1;
  return (int)(b > cap ? cap : b);
}

}  // namespace

DVAE_API int dvae_latent_fwd(const float* style, const float* content, const float* eps_c, const float* eps_s,
                             float* z, float* q_mu, float* q_lv, float* s_mu, float* s_lv, int Bh, int S, int Cn,
                             void* stream) {
  if (!style || !content || !eps_s || !z || !q_mu || !q_lv || !s_mu || !s_lv || Bh < 1 || S < 1 || Cn < 1)
    return DVAE_EINVAL;
  const int total = 2 * Bh * (S + Cn);
  hipLaunchKernelGGL(latent_fwd_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, style, content,
                     eps_c, eps_s, z, q_mu, q_lv, s_mu, s_lv, Bh, S, Cn);
  return dvae_check_launch();
}

DVAE_API int dvae_latent_bwd(const float* style, const float* content, const float* eps_c, const float* eps_s,
                             const float* dz, const float* dq_mu, const float* dq_lv, const float* ds_mu,
                             const float* ds_lv, float* dstyle, float* dcontent, int Bh, int S, int Cn,
                             void* stream) {
  if (!style || !content || !eps_s || !dstyle || !dcontent || Bh < 1 || S < 1 || Cn < 1) return DVAE_EINVAL;
  const int total = 2 * Bh * (S + Cn);
  hipLaunchKernelGGL(latent_bwd_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, style, content,
                     eps_c, eps_s, dz, dq_mu, dq_lv, ds_mu, ds_lv, dstyle, dcontent, Bh, S, Cn);
  return dvae_check_launch();
}

DVAE_API int dvae_kl_fwd(const float* mu, const float* lv, float* out, int64_t n, float scale, void* stream) {
  if (!mu || !lv || !out || n < 1) return DVAE_EINVAL;
  hipLaunchKernelGGL(kl_fwd_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, mu, lv, out, n, scale);
  return dvae_check_launch();
}

DVAE_API int dvae_kl_bwd(const float* mu, const float* lv, const float* gout, float* dmu, float* dlv, int64_t n,
                         float scale, void* stream) {
  if (!mu || !lv || !gout || !dmu || !dlv || n < 1) return DVAE_EINVAL;
  hipLaunchKernelGGL(kl_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, mu, lv,
                     gout, dmu, dlv, n, scale);
  return dvae_check_launch();
}

DVAE_API int64_t dvae_l1_ws_bytes(int64_t n) {
  (void)n;
  return (int64_t)L1_BLOCKS * sizeof(double);
}

DVAE_API int dvae_l1_sum_fwd(const float* x, const float* y, float* out, void* ws, int64_t n, float scale,
                             void* stream) {
  if (!x || !y || !out || !ws || n < 1) return DVAE_EINVAL;
  if ((((uintptr_t)x) | ((uintptr_t)y)) & 15) return DVAE_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  const int blocks = nblk(n / 4 + 1, 256, L1_BLOCKS);
  hipLaunchKernelGGL(l1_partial_kernel, dim3(blocks), dim3(256), 0, s, x, y, (double*)ws, n);
  hipLaunchKernelGGL(l1_final_kernel, dim3(1), dim3(64), 0, s, (const double*)ws, out, blocks, scale);
  return dvae_check_launch();
}

DVAE_API int dvae_l1_sum_bwd(const float* x, const float* y, const float* gout, float* dy, int64_t n, float scale,
                             void* stream) {
  if (!x || !y || !gout || !dy || n < 1) return DVAE_EINVAL;
  hipLaunchKernelGGL(l1_bwd_kernel, dim3(nblk(n)), dim3(256), 0, (hipStream_t)stream, x, y, gout, dy, n, scale);
  return dvae_check_launch();
}


namespace {
int fill_loss_args(LossArgs& a, const dvae_loss_desc_t* d) {
  if (!d || d->n < 1 || d->nq < 1 || d->ns < 1) return DVAE_EINVAL;
  const void* ps[] = {d->x1, d->x2, d->recon1, d->recon2, d->recon1_hat, d->recon2_hat, d->q1_mu, d->q1_lv, d->q2_mu, d->q2_lv,
                      d->s_mu, d->s_lv};
  for (const void* p : ps)
    if (!p) return DVAE_EINVAL;
  for (int i = 0; i < 6; ++i)
    if (((uintptr_t)ps[i]) & 15) return DVAE_EINVAL;
  a.x[0] = d->x1; a.x[1] = d->x2;
  a.r[0] = d->recon1; a.r[1] = d->recon2; a.r[2] = d->recon1_hat; a.r[3] = d->recon2_hat;
  a.qmu[0] = d->q1_mu; a.qlv[0] = d->q1_lv; a.qmu[1] = d->q2_mu; a.qlv[1] = d->q2_lv;
  a.smu = d->s_mu; a.slv = d->s_lv;
  a.n = d->n; a.nq = d->nq; a.ns = d->ns;
  a.l1_scale = d->l1_scale; a.kl_scale = d->kl_scale; a.style_scale = d->style_scale;
  a.mse_cof = d->mse_cof; a.kl_cof = d->kl_cof;
  return DVAE_OK;
}
}  // namespace

DVAE_API int64_t dvae_loss_ws_bytes(int64_t n) {
  (void)n;
  return (int64_t)L1_BLOCKS * 4 * sizeof(double);
}

DVAE_API int dvae_loss_fwd(const dvae_loss_desc_t* desc, float* out8, void* ws, void* stream) {
  LossArgs a;
  int rc = fill_loss_args(a, desc);
  if (rc) return rc;
  if (!out8 || !ws) return DVAE_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  const int blocks = nblk(a.n / 4 + 1, 256, L1_BLOCKS);
  hipLaunchKernelGGL(loss_partial_kernel, dim3(blocks), dim3(256), 0, s, a, (double*)ws);
  hipLaunchKernelGGL(loss_final_kernel, dim3(1), dim3(256), 0, s, a, (const double*)ws, blocks, out8);
  return dvae_check_launch();
}

DVAE_API int dvae_loss_bwd(const dvae_loss_desc_t* desc, const float* g8, float* d_recon1, float* d_recon2,
                           float* d_recon1_hat, float* d_recon2_hat, float* d_q1_mu, float* d_q1_lv, float* d_q2_mu,
                           float* d_q2_lv, float* d_s_mu, float* d_s_lv, void* stream) {
  LossArgs a;
  int rc = fill_loss_args(a, desc);
  if (rc) return rc;
  if (!g8 || (!d_q1_mu != !d_q1_lv) || (!d_q2_mu != !d_q2_lv) || (!d_s_mu != !d_s_lv)) return DVAE_EINVAL;
  const float* outs[] = {d_recon1, d_recon2, d_recon1_hat, d_recon2_hat};
  for (const float* p : outs)
    if (p && (((uintptr_t)p) & 15)) return DVAE_EINVAL;
  const int blocks = nblk(a.n / 4 + 1, 256, 1024);
  hipLaunchKernelGGL(loss_bwd_kernel, dim3(blocks + 1), dim3(256), 0, (hipStream_t)stream, a, g8, d_recon1, d_recon2,
                     d_recon1_hat, d_recon2_hat, d_q1_mu, d_q1_lv, d_q2_mu, d_q2_lv, d_s_mu, d_s_lv, blocks);
  return dvae_check_launch();
}

DVAE_API int dvae_adam_flat(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1,
                            float beta2, float eps, float grad_scale, int step, void* stream) {
  if (!p || !g || !m || !v || n < 1 || step < 1) return DVAE_EINVAL;
  if ((((uintptr_t)p) | ((uintptr_t)g) | ((uintptr_t)m) | ((uintptr_t)v)) & 15) return DVAE_EINVAL;
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  hipLaunchKernelGGL(adam_kernel, dim3(nblk(n / 4 + 1, 256, 4096)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n,
                     lr, beta1, beta2, eps, grad_scale, (float)bc1, (float)sqrt(bc2));
  return dvae_check_launch();
}

DVAE_API int dvae_adam_flat_dev(float* p, float* g, float* m, float* v, int64_t n, float beta1, float beta2, float eps,
                                float* state, const unsigned* skip_if_nonzero, const dvae_ranges_t* clear, int tick,
                                void* stream) {
  if (!p || !g || !m || !v || !state || n < 4 || (n & 3)) return DVAE_EINVAL;
  if ((((uintptr_t)p) | ((uintptr_t)g) | ((uintptr_t)m) | ((uintptr_t)v)) & 15) return DVAE_EINVAL;
  AdamClear c{};
  if (clear) {
    if (clear->n < 0 || clear->n > 8) return DVAE_EINVAL;
    c.n = clear->n;
    for (int r = 0; r < c.n; ++r) {
      if (clear->lo[r] < 0 || clear->hi[r] > n || clear->lo[r] > clear->hi[r] || ((clear->lo[r] | clear->hi[r]) & 3))
        return DVAE_EINVAL;
      c.lo[r] = clear->lo[r];
      c.hi[r] = clear->hi[r];
    }
  }
  hipStream_t s = (hipStream_t)stream;
  if (tick) hipLaunchKernelGGL(adam_tick_kernel, dim3(1), dim3(64), 0, s, state, beta1, beta2, skip_if_nonzero);
#ifndef DVAE_ADAM_U
#define DVAE_ADAM_U 2
#endif
  constexpr int U = DVAE_ADAM_U;
  const int64_t n4 = n >> 2;
  hipLaunchKernelGGL(adam_dev_kernel<U>, dim3(nblk((n4 + U - 1) / U, 256, 2048)), dim3(256), 0, s, p, g, m, v, n4, beta1,
                     beta2, eps, state, skip_if_nonzero, c);
  return dvae_check_launch();
}

// x[0, n) = 0 with plain stores, in a launch of its own right in front of the launches that accumulate into x atomically
// (gradients: FlatAdam.zero_grad; split-k outputs of the Linear layers)
__global__ __launch_bounds__(256) void zero_kernel(float* __restrict__ x, int64_t n) {
  const int64_t n4 = n >> 2;
  const f32x4 z = {0.f, 0.f, 0.f, 0.f};
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256)
    *reinterpret_cast<f32x4*>(x + 4 * i) = z;
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) x[(n4 << 2) + threadIdx.x] = 0.f;
}
DVAE_API int dvae_zero_f32(float* x, int64_t n, void* stream) {
  if (!x || n < 1 || (((uintptr_t)x) & 15)) return DVAE_EINVAL;
  hipLaunchKernelGGL(zero_kernel, dim3(nblk(n / 4 + 1, 256, 2048)), dim3(256), 0, (hipStream_t)stream, x, n);
  return dvae_check_launch();
}

// out = a + b (+ c): the sum of the gradients that reach a tensor with several consumers (ops.FanoutFn)
__global__ __launch_bounds__(256) void sum_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                  const float* __restrict__ c, float* __restrict__ out, int64_t n4) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    f32x4 v = *reinterpret_cast<const f32x4*>(a + 4 * i) + *reinterpret_cast<const f32x4*>(b + 4 * i);
    if (c) v += *reinterpret_cast<const f32x4*>(c + 4 * i);
    *reinterpret_cast<f32x4*>(out + 4 * i) = v;
  }
}
DVAE_API int dvae_sum_f32(const float* a, const float* b, const float* c, float* out, int64_t n, void* stream) {
  if (!a || !b || !out || n < 4 || (n & 3)) return DVAE_EINVAL;
  if ((((uintptr_t)a) | ((uintptr_t)b) | ((uintptr_t)c) | ((uintptr_t)out)) & 15) return DVAE_EINVAL;
  hipLaunchKernelGGL(sum_kernel, dim3(nblk(n / 4, 256, 2048)), dim3(256), 0, (hipStream_t)stream, a, b, c, out, n >> 2);
  return dvae_check_launch();
}

DVAE_API int dvae_gather_crop(const float* mels, const int* lens, const int* utt, const int* off, float* out, int n,
                              int C, int T, int Lmax, void* stream) {
  if (!mels || !lens || !utt || !off || !out || n < 1 || C < 1 || T < 1 || Lmax < 1) return DVAE_EINVAL;
  hipLaunchKernelGGL(gather_crop_kernel, dim3(nblk((int64_t)n * C * T)), dim3(256), 0, (hipStream_t)stream, mels, lens,
                     utt, off, out, n, C, T, Lmax);
  return dvae_check_launch();
}

DVAE_API int dvae_mel_to_chunks(const float* mel, float* out, int C, int L, int T, int n, void* stream) {
  if (!mel || !out || C < 1 || L < 0 || T < 1 || n < 1) return DVAE_EINVAL;
  hipLaunchKernelGGL(mel_to_chunks_kernel, dim3(nblk((int64_t)n * C * T)), dim3(256), 0, (hipStream_t)stream, mel, out,
                     C, L, T, n);
  return dvae_check_launch();
}
DVAE_API int dvae_chunks_to_mel(const float* in, float* out, int n, int C, int T, float lo, float hi, int clamp,
                                void* stream) {
  if (!in || !out || n < 1 || C < 1 || T < 1) return DVAE_EINVAL;
  hipLaunchKernelGGL(chunks_to_mel_kernel, dim3(nblk((int64_t)n * C * T)), dim3(256), 0, (hipStream_t)stream, in, out,
                     n, C, T, lo, hi, clamp);
  return dvae_check_launch();
}
DVAE_API int dvae_conversion_latents(const float* src_style, const float* src_content, const float* trg_style,
                                     float* z_src, float* z_conv, int n, int m, int S, int Cn, void* stream) {
  if (!src_style || !src_content || !trg_style || !z_src || !z_conv || n < 1 || m < 1 || S < 1 || Cn < 1)
    return DVAE_EINVAL;
  const int total = n * (S + Cn);
  hipLaunchKernelGGL(conversion_latents_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, src_style,
                     src_content, trg_style, z_src, z_conv, n, m, S, Cn);
  return dvae_check_launch();
}
DVAE_API int dvae_mul_div(const float* a, const float* b, const float* c, float* out, int64_t n, void* stream) {
  if (!a || !b || !c || !out || n < 1) return DVAE_EINVAL;
  hipLaunchKernelGGL(mul_div_kernel, dim3(nblk(n)), dim3(256), 0, (hipStream_t)stream, a, b, c, out, n);
  return dvae_check_launch();
}

DVAE_API int dvae_mel_to_frames(const float* x1, const float* x2, void* X, int Bh, int C, int T, int out_bf16,
                                void* stream) {
  if (!x1 || !X || Bh < 1 || C < 1 || T < 1) return DVAE_EINVAL;
  const int N = x2 ? 2 * Bh : Bh;
  dim3 grid((T + 31) / 32, (C + 31) / 32, N);
  if (out_bf16) hipLaunchKernelGGL(mel_to_frames_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, x1, x2, X, Bh, C, T, N);
  else hipLaunchKernelGGL(mel_to_frames_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, x1, x2, X, Bh, C, T, N);
  return dvae_check_launch();
}

DVAE_API int dvae_frames_to_mel(const float* X, float* out, int N, int C, int T, void* stream) {
  if (!X || !out || N < 1 || C < 1 || T < 1) return DVAE_EINVAL;
  dim3 grid((T + 31) / 32, (C + 31) / 32, N);
  hipLaunchKernelGGL(frames_to_mel_kernel, grid, dim3(256), 0, (hipStream_t)stream, X, out, N, C, T);
  return dvae_check_launch();
}

DVAE_API int dvae_permute_102(const float* in, float* out, int A, int B, int C, void* stream) {
  if (!in || !out || A < 1 || B < 1 || C < 4 || (C & 3)) return DVAE_EINVAL;
  const int64_t total = (int64_t)A * B * (C / 4);
  hipLaunchKernelGGL(permute_102_kernel, dim3(nblk(total)), dim3(256), 0, (hipStream_t)stream, in, out, A, B, C / 4);
  return dvae_check_launch();
}

DVAE_API int dvae_colsum_add(const void* X, float* out1, float* out2, int R, int C, int64_t ld, int x_bf16, void* stream) {
  if (!X || !out1 || R < 1 || C < 1) return DVAE_EINVAL;
  if ((ld & 3) || (((uintptr_t)X) & (x_bf16 ? 7 : 15))) return DVAE_EINVAL;
  // (fewer rows per workgroup for narrow matrices -- more workgroups -- was tried: the extra same-address atomics
  // cost more than the parallelism gains, 0.43 -> 1.0 ms per step)
  const int cb = (C + 255) / 256;
  // Rows per workgroup: every workgroup ends in one atomic per column, so few, long workgroups win (measured, us at
  // 128 / 256 / 512 / 1024 rows: [16384 x 512] fp32 18.0 / 9.6 / 8.3 / 10.6; [65536 x 512] bf16 57 / 35 / 23 / 22;
  // [16384 x 4096] fp32 54 / 52 / 48 / 41): 512, and 1024 once there are >= 512 workgroups even so
  int rows_pb = ((int64_t)R * cb >= (int64_t)512 * 1024) ? 1024 : 512;
  if (dvae_dev_knob("DVAE_COLSUM_ROWS", 0) > 0) rows_pb = dvae_dev_knob("DVAE_COLSUM_ROWS", 0);
  if (g_dvae_deterministic) rows_pb = R;      // one workgroup per column block walks all rows: one add per column
  dim3 grid(cb, (R + rows_pb - 1) / rows_pb);
  if (x_bf16) hipLaunchKernelGGL(colsum_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, X, out1, out2, R, C, ld, rows_pb);
  else hipLaunchKernelGGL(colsum_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, X, out1, out2, R, C, ld, rows_pb);
  return dvae_check_launch();
}

// ---------------------------------------------------------------------------------------------------------------------
// k-split without atomics (round 6): the partial products a contraction stored into slabs (dvae_gemm_f32_slabs ...) are
// added to the result in the FIXED order ks = 1, 2, ...: every element has one writer and one summation order, so the
// result is run-to-run bit-identical.  HBM-bound: (nslab + 2) * 4 bytes per element.
// v + slab 0 + slab 1 + ... in THAT order; eight 16-byte loads in flight per thread (the trip count is a run-time value: without
// the explicit window the loads of a thread go out one behind the other)
__device__ __forceinline__ f32x4 slab_add(f32x4 v, const float* __restrict__ p, int64_t stride, int nslab) {
  int k = 0;
  for (; k + 8 <= nslab; k += 8) {
    f32x4 w[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) w[j] = *reinterpret_cast<const f32x4*>(p + (int64_t)(k + j) * stride);
#pragma unroll
    for (int j = 0; j < 8; ++j) v += w[j];
  }
  if (k + 4 <= nslab) {
    f32x4 w[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) w[j] = *reinterpret_cast<const f32x4*>(p + (int64_t)(k + j) * stride);
#pragma unroll
    for (int j = 0; j < 4; ++j) v += w[j];
    k += 4;
  }
  for (; k < nslab; ++k) v += *reinterpret_cast<const f32x4*>(p + (int64_t)k * stride);
  return v;
}
__global__ __launch_bounds__(256) void slab_sum_kernel(float* __restrict__ C, const float* __restrict__ slab, int64_t stride,
                                                       int nslab, int64_t n4, int act, int accumulate) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (accumulate) v = *reinterpret_cast<const f32x4*>(C + 4 * i);
    v = slab_add(v, slab + 4 * i, stride, nslab);
    if (act != DVAE_ACT_NONE) {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = act_apply(v[e], act);
    }
    *reinterpret_cast<f32x4*>(C + 4 * i) = v;
  }
}
// C[i] = act((accumulate ? C[i] : 0) + sum_k slab[k * slab_stride + i]), i < n (n, slab_stride multiples of 4; 16-byte aligned)
DVAE_API int dvae_slab_sum(float* C, const float* slab, int64_t slab_stride, int nslab, int64_t n, int act, int accumulate,
                           void* stream) {
  if (!C || n < 4 || (n & 3) || nslab < 0 || (nslab > 0 && (!slab || (slab_stride & 3) || (((uintptr_t)slab) & 15))) ||
      (((uintptr_t)C) & 15) || (nslab == 0 && !accumulate))
    return DVAE_EINVAL;
  if (nslab == 0 && act == DVAE_ACT_NONE) return DVAE_OK;
  hipLaunchKernelGGL(slab_sum_kernel, dim3(nblk(n / 4, 256, 2048)), dim3(256), 0, (hipStream_t)stream, C, slab, slab_stride,
                     nslab, n >> 2, act, accumulate);
  return dvae_check_launch();
}

// the same for MANY results in one launch (the weight gradients of a step, in front of the Adam launch or of a bucket's
// collective): entry e adds its nslab slabs to its n elements.  The table travels as a kernel argument (<= 64 entries).
struct SlabTable {
  dvae_slab_desc_t e[DVAE_SLAB_FOLD_MAX];
};
__global__ __launch_bounds__(256) void slab_fold_kernel(const SlabTable t) {
  const dvae_slab_desc_t d = t.e[blockIdx.y];
  const int64_t n4 = d.n >> 2;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    f32x4 v = *reinterpret_cast<const f32x4*>(d.c + 4 * i);
    v = slab_add(v, d.slab + 4 * i, d.slab_stride, d.nslab);
    *reinterpret_cast<f32x4*>(d.c + 4 * i) = v;
  }
}
DVAE_API int dvae_slab_fold(const dvae_slab_desc_t* descs, int n_entries, void* stream) {
  if (!descs || n_entries < 0) return DVAE_EINVAL;
  for (int base = 0; base < n_entries; base += DVAE_SLAB_FOLD_MAX) {
    SlabTable t{};
    const int m = n_entries - base < DVAE_SLAB_FOLD_MAX ? n_entries - base : DVAE_SLAB_FOLD_MAX;
    int64_t nmax = 0;
    for (int i = 0; i < m; ++i) {
      const dvae_slab_desc_t& d = descs[base + i];
      if (!d.c || !d.slab || d.n < 4 || (d.n & 3) || (d.slab_stride & 3) || d.nslab < 1 ||
          ((((uintptr_t)d.c) | ((uintptr_t)d.slab)) & 15))
        return DVAE_EINVAL;
      t.e[i] = d;
      nmax = d.n > nmax ? d.n : nmax;
    }
    hipLaunchKernelGGL(slab_fold_kernel, dim3(nblk(nmax / 4, 256, 512), m), dim3(256), 0, (hipStream_t)stream, t);
  }
  return dvae_check_launch();
}

// ---------------------------------------------------------------------------------------------------------------------
// Column sums WITHOUT atomics: every workgroup stores the partial sums of its row block, the LAST workgroup of a column
// block to arrive (a ticket counter per column block, agent-scope release / acquire around it) adds them up in row-block
// order and is the one writer of out[c].  ws: [column blocks] counters (zero when the call starts, left zero) + the partials.
template <bool XB16>
__global__ __launch_bounds__(256) void colsum_ws_kernel(const void* __restrict__ Xv, float* __restrict__ o1,
                                                        float* __restrict__ o2, int R, int C, int64_t ld, int rows_pb,
                                                        unsigned* __restrict__ cnt, float* __restrict__ part) {
  using elem_t = typename std::conditional<XB16, __bf16, float>::type;
  const elem_t* __restrict__ X = reinterpret_cast<const elem_t*>(Xv);
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int c = (blockIdx.x * 64 + cl) * 4;
  const int r0 = blockIdx.y * rows_pb, r1 = min(R, r0 + rows_pb);
  const int Cp = gridDim.x * 256;                    // padded row length of the partials
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  if (c + 3 < C) {
    const elem_t* __restrict__ px = X + (int64_t)(r0 + rl) * ld + c;
    const int n = (r1 - r0 - rl + 3) >> 2;
    int i = 0;
    for (; i + 8 <= n; i += 8) {
      f32x4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = ld4<XB16>(px + (int64_t)(i + u) * 4 * ld, 0);
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; i < n; ++i) s += ld4<XB16>(px + (int64_t)i * 4 * ld, 0);
  } else if (c < C) {
    for (int r = r0 + rl; r < r1; r += 4)
      for (int k = 0; k < 4 && c + k < C; ++k) s[k] += (float)X[(int64_t)r * ld + c + k];
  }
  __shared__ f32x4 red[4][64];
  __shared__ int last;
  red[rl][cl] = s;
  __syncthreads();
  if (rl == 0) {
    const f32x4 tot = (red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl]);
    *reinterpret_cast<f32x4*>(part + (int64_t)blockIdx.y * Cp + (blockIdx.x * 64 + cl) * 4) = tot;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned tk = __hip_atomic_fetch_add(cnt + blockIdx.x, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    last = (tk == gridDim.y - 1);
    if (last) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  }
  __syncthreads();
  if (!last) return;
  // wave rl adds the row blocks rl, rl + 4, ... in that order, then the four waves' sums are combined in a fixed tree: the
  // same sum in every run
  f32x4 ps = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
  for (unsigned b = rl; b < gridDim.y; b += 4)
    ps += *reinterpret_cast<const f32x4*>(part + (int64_t)b * Cp + (blockIdx.x * 64 + cl) * 4);
  red[rl][cl] = ps;
  __syncthreads();
  if (rl == 0 && c < C) {
    const f32x4 tot = (red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl]);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (c + k >= C) break;
      o1[c + k] += tot[k];
      if (o2) o2[c + k] += tot[k];
    }
  }
  if (threadIdx.x == 0) __hip_atomic_store(cnt + blockIdx.x, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

static inline int colsum_rows_pb(int R, int C) {
  const int cb = (C + 255) / 256;
  return ((int64_t)R * cb >= (int64_t)512 * 1024) ? 1024 : 512;
}
DVAE_API int64_t dvae_colsum_ws_bytes(int R, int C) {
  if (R < 1 || C < 1) return 0;
  const int cb = (C + 255) / 256, rows_pb = colsum_rows_pb(R, C);
  return 4096 + (int64_t)((R + rows_pb - 1) / rows_pb) * cb * 256 * 4;
}
// out1[c] (and out2[c]) += sum_r X[r][c], deterministic (see above).  ws: >= dvae_colsum_ws_bytes(R, C) bytes, 16-byte aligned,
// its first 4096 bytes ZERO before the first call (the counters; every call leaves them zero); one ws per stream in flight
DVAE_API int dvae_colsum_add_ws(const void* X, float* out1, float* out2, int R, int C, int64_t ld, int x_bf16, void* ws,
                                void* stream) {
  if (!X || !out1 || !ws || R < 1 || C < 1 || C > 256 * 1024) return DVAE_EINVAL;
  if ((ld & 3) || (((uintptr_t)X) & (x_bf16 ? 7 : 15)) || (((uintptr_t)ws) & 15)) return DVAE_EINVAL;
  const int cb = (C + 255) / 256, rows_pb = colsum_rows_pb(R, C);
  dim3 grid(cb, (R + rows_pb - 1) / rows_pb);
  unsigned* cnt = (unsigned*)ws;
  float* part = (float*)((char*)ws + 4096);
  if (x_bf16) hipLaunchKernelGGL(colsum_ws_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, X, out1, out2, R, C, ld, rows_pb, cnt, part);
  else hipLaunchKernelGGL(colsum_ws_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, X, out1, out2, R, C, ld, rows_pb, cnt, part);
  return dvae_check_launch();
}

DVAE_API int dvae_transpose(const float* in, float* out, int R, int C, void* stream) {
  if (!in || !out || R < 1 || C < 1) return DVAE_EINVAL;
  dim3 grid((C + 31) / 32, (R + 31) / 32);
  hipLaunchKernelGGL(transpose_kernel, grid, dim3(256), 0, (hipStream_t)stream, in, out, R, C);
  return dvae_check_launch();
}

DVAE_API int dvae_act_bwd(const float* dZ, const float* Z, float* dU, int64_t n, int act, void* stream) {
  if (!dZ || !Z || !dU || n < 1) return DVAE_EINVAL;
  hipLaunchKernelGGL(act_bwd_kernel, dim3(nblk(n)), dim3(256), 0, (hipStream_t)stream, dZ, Z, dU, n, act);
  return dvae_check_launch();
}

DVAE_API int dvae_act_fwd(float* Y, int64_t n, int act, void* stream) {
  if (!Y || n < 1) return DVAE_EINVAL;
  hipLaunchKernelGGL(act_fwd_kernel, dim3(nblk(n)), dim3(256), 0, (hipStream_t)stream, Y, n, act);
  return dvae_check_launch();
}

DVAE_API int dvae_conv_pack_w(const float* W, float* Wp, int Cout, int Cin, void* stream) {
  if (!W || !Wp || Cout < 1 || Cin < 1) return DVAE_EINVAL;
  const int64_t cc = (int64_t)Cout * Cin;
  hipLaunchKernelGGL(conv_pack_kernel, dim3(nblk(cc)), dim3(256), 0, (hipStream_t)stream, W, Wp, cc);
  return dvae_check_launch();
}

DVAE_API int dvae_conv_pack_wt(const float* W, float* Wpt, int Cout, int Cin, void* stream) {
  if (!W || !Wpt || Cout < 1 || Cin < 1) return DVAE_EINVAL;
  dim3 grid((Cin + 31) / 32, (Cout + 31) / 32);
  hipLaunchKernelGGL(conv_pack_t_kernel, grid, dim3(256), 0, (hipStream_t)stream, W, Wpt, Cout, Cin);
  return dvae_check_launch();
}

DVAE_API int dvae_conv_unpack_add_w(const float* dWp, float* dW, int Cout, int Cin, void* stream) {
  if (!dWp || !dW || Cout < 1 || Cin < 1) return DVAE_EINVAL;
  const int64_t cc = (int64_t)Cout * Cin;
  hipLaunchKernelGGL(conv_unpack_add_kernel, dim3(nblk(cc)), dim3(256), 0, (hipStream_t)stream, dWp, dW, cc);
  return dvae_check_launch();
}
