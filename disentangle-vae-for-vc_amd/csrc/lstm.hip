// Frame-major LSTM recurrence for gfx950: one launch per frame (a kernel boundary, ~1.5 us, is
// cheaper on MI355X than any in-kernel grid barrier), all launches of a sequence issued from one
// C call so the host cost is a tight C loop (and a hipGraph captures it verbatim).
//
// Forward step  :  G = Xproj[t] + H[t-1] * W_hh^T ;  i,f,o = sigmoid, g = tanh ; c = f*c' + i*g ; h = o*tanh(c)
// Backward step :  dH = dHout[t] + dG[t+1] * W_hh ; gate derivatives ; dG[t] ; dC carry
//
// Work split (both directions): a workgroup owns 16 hidden units x (16*MT) mel segments and has
// four waves.  Forward: wave g computes gate g's 16x16 tiles over the full K = H with
// v_mfma_f32_16x16x4_f32; the four gate tiles meet in LDS for the fused pointwise update.
// Backward: the contraction runs over K = 4H, wave w takes the quarter [w*H,(w+1)*H) and the four
// partial tiles are summed in LDS before the fused gate-derivative epilogue.
// Operands are k-contiguous in memory, so each lane fetches float4 fragments straight from
// L2 (W_hh is re-read every frame and stays L2/MALL resident; the blockIdx -> tile map keeps all
// row-tiles of one weight slice on one XCD).  The k order inside a 16-chunk is permuted
// (lane group q takes k = 4q..4q+3) identically for A and B, which leaves the dot product intact.
#include "common.h"

namespace {

struct StepDir {
  float* gates;         // [T,N,4H]
  const float* w;       // fwd [4H,H] ; bwd W_hh^T [H,4H]
  float* h_out;         // [T,N,ldh] (column offset already applied)
  float* c_all;         // [T,N,H]
  const float* dh_out;  // [T,N,ldh]
  float* dgates;        // [T,N,4H]
  float* dc;            // [N,H]
  int reverse;
};
struct StepArgs {
  StepDir d[2];
  int T, N, H;
  int64_t ldh;
};

// XCD-aware decode of the linear block id into (j-block, m-block): consecutive ids go to
// different XCDs (round-robin dispatch), ids equal mod 8 share one.  Put all m-blocks of a
// j-block on one XCD so a W_hh slice lives in exactly one L2.
__device__ __forceinline__ void decode_block(int bid, int n_j, int n_m, int& jb, int& mb) {
  if ((n_j & 7) == 0) {
    const int x = bid & 7;        // XCD label
    const int q = bid >> 3;       // index inside that XCD
    mb = q % n_m;
    jb = (q / n_m) * 8 + x;
  } else {
    mb = bid % n_m;
    jb = bid / n_m;
  }
}

template <int MT>
__global__ __launch_bounds__(256) void lstm_step_fwd_kernel(const StepArgs a, int step, int n_j, int n_m) {
  const StepDir& d = a.d[blockIdx.z];
  const int H = a.H, N = a.N;
  const int t = d.reverse ? (a.T - 1 - step) : step;
  const int tp = d.reverse ? t + 1 : t - 1;
  int jb, mb;
  decode_block(blockIdx.x, n_j, n_m, jb, mb);
  const int j0 = jb * 16, m0 = mb * 16 * MT;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, kq = lane >> 4;

  __shared__ float sm[4][MT * 16][17];

  f32x4 acc[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};

  if (step > 0) {
    const float* __restrict__ hp = d.h_out + (int64_t)tp * N * a.ldh;
    const float* __restrict__ wrow = d.w + ((int64_t)wave * H + j0 + r) * H + 4 * kq;
    const float* arow[MT];
    bool aok[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int row = m0 + mt * 16 + r;
      aok[mt] = row < N;
      arow[mt] = hp + (int64_t)(aok[mt] ? row : 0) * a.ldh + 4 * kq;
    }
    // two register buffers of 32 k each; H is a multiple of 64
    f32x4 b0[2], b1[2], a0[MT][2], a1[MT][2];
    auto ld = [&](f32x4 (&bb)[2], f32x4 (&aa)[MT][2], int k0) {
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        bb[c] = *reinterpret_cast<const f32x4*>(wrow + k0 + 16 * c);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
          aa[mt][c] = aok[mt] ? *reinterpret_cast<const f32x4*>(arow[mt] + k0 + 16 * c) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
    };
    auto mm = [&](f32x4 (&bb)[2], f32x4 (&aa)[MT][2]) {
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
            acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(aa[mt][c][e], bb[c][e], acc[mt], 0, 0, 0);
    };
    ld(b0, a0, 0);
    for (int k0 = 0; k0 < H; k0 += 64) {
      ld(b1, a1, k0 + 32);
      mm(b0, a0);
      if (k0 + 64 < H) ld(b0, a0, k0 + 64);
      mm(b1, a1);
    }
  }
  // C/D map of the 16x16 tile: col = lane&15, row = (lane>>4)*4 + reg
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int e = 0; e < 4; ++e) sm[wave][mt * 16 + kq * 4 + e][r] = acc[mt][e];
  __syncthreads();

  float* __restrict__ G = d.gates + (int64_t)t * N * 4 * H;
  const float* __restrict__ cprev = d.c_all + (int64_t)tp * N * H;
  float* __restrict__ cout = d.c_all + (int64_t)t * N * H;
  float* __restrict__ hout = d.h_out + (int64_t)t * N * a.ldh;
#pragma unroll
  for (int e = 0; e < MT; ++e) {
    const int idx = tid + 256 * e;
    const int row = idx >> 4, col = idx & 15;
    const int n = m0 + row;
    if (n >= N) continue;
    const int j = j0 + col;
    float* g = G + (int64_t)n * 4 * H + j;
    const float gi = sigmoidf_(sm[0][row][col] + g[0]);
    const float gf = sigmoidf_(sm[1][row][col] + g[H]);
    const float gg = tanhf(sm[2][row][col] + g[2 * H]);
    const float go = sigmoidf_(sm[3][row][col] + g[3 * H]);
    const float cp = (step > 0) ? cprev[(int64_t)n * H + j] : 0.f;
    const float c = gf * cp + gi * gg;
    g[0] = gi;
    g[H] = gf;
    g[2 * H] = gg;
    g[3 * H] = go;
    cout[(int64_t)n * H + j] = c;
    hout[(int64_t)n * a.ldh + j] = go * tanhf(c);
  }
}

template <int MT>
__global__ __launch_bounds__(256) void lstm_step_bwd_kernel(const StepArgs a, int step, int n_j, int n_m) {
  // step counts backward-time iterations: step 0 handles the LAST frame of the forward recurrence
  const StepDir& d = a.d[blockIdx.z];
  const int H = a.H, N = a.N;
  const int fstep = a.T - 1 - step;                       // position in forward-recurrence order
  const int t = d.reverse ? (a.T - 1 - fstep) : fstep;     // frame index
  const int tn = d.reverse ? t - 1 : t + 1;                // frame processed AFTER t in the forward recurrence
  const int tp = d.reverse ? t + 1 : t - 1;                // frame processed BEFORE t
  int jb, mb;
  decode_block(blockIdx.x, n_j, n_m, jb, mb);
  const int j0 = jb * 16, m0 = mb * 16 * MT;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, kq = lane >> 4;

  __shared__ float sm[4][MT * 16][17];

  f32x4 acc[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};

  if (step > 0) {
    // dHrec[n, j] = sum_k dG[tn][n, k] * W_hh[k, j] ; this wave: k in [wave*H, (wave+1)*H)
    const float* __restrict__ dgn = d.dgates + (int64_t)tn * N * 4 * H + (int64_t)wave * H + 4 * kq;
    const float* __restrict__ wrow = d.w + (int64_t)(j0 + r) * 4 * H + (int64_t)wave * H + 4 * kq;
    const float* arow[MT];
    bool aok[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int row = m0 + mt * 16 + r;
      aok[mt] = row < N;
      arow[mt] = dgn + (int64_t)(aok[mt] ? row : 0) * 4 * H;
    }
    f32x4 b0[2], b1[2], a0[MT][2], a1[MT][2];
    auto ld = [&](f32x4 (&bb)[2], f32x4 (&aa)[MT][2], int k0) {
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        bb[c] = *reinterpret_cast<const f32x4*>(wrow + k0 + 16 * c);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
          aa[mt][c] = aok[mt] ? *reinterpret_cast<const f32x4*>(arow[mt] + k0 + 16 * c) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
    };
    auto mm = [&](f32x4 (&bb)[2], f32x4 (&aa)[MT][2]) {
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
            acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(aa[mt][c][e], bb[c][e], acc[mt], 0, 0, 0);
    };
    ld(b0, a0, 0);
    for (int k0 = 0; k0 < H; k0 += 64) {
      ld(b1, a1, k0 + 32);
      mm(b0, a0);
      if (k0 + 64 < H) ld(b0, a0, k0 + 64);
      mm(b1, a1);
    }
  }
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int e = 0; e < 4; ++e) sm[wave][mt * 16 + kq * 4 + e][r] = acc[mt][e];
  __syncthreads();

  const float* __restrict__ G = d.gates + (int64_t)t * N * 4 * H;
  float* __restrict__ dG = d.dgates + (int64_t)t * N * 4 * H;
  const float* __restrict__ ccur = d.c_all + (int64_t)t * N * H;
  const float* __restrict__ cprev = d.c_all + (int64_t)tp * N * H;
  const float* __restrict__ dho = d.dh_out + (int64_t)t * N * a.ldh;
#pragma unroll
  for (int e = 0; e < MT; ++e) {
    const int idx = tid + 256 * e;
    const int row = idx >> 4, col = idx & 15;
    const int n = m0 + row;
    if (n >= N) continue;
    const int j = j0 + col;
    const float dh = dho[(int64_t)n * a.ldh + j] + sm[0][row][col] + sm[1][row][col] + sm[2][row][col] + sm[3][row][col];
    const float* g = G + (int64_t)n * 4 * H + j;
    const float gi = g[0], gf = g[H], gg = g[2 * H], go = g[3 * H];
    const float c = ccur[(int64_t)n * H + j];
    const float tc = tanhf(c);
    const float cp = (fstep > 0) ? cprev[(int64_t)n * H + j] : 0.f;
    const float dcar = (step > 0) ? d.dc[(int64_t)n * H + j] : 0.f;
    const float dc = dcar + dh * go * (1.f - tc * tc);
    float* o = dG + (int64_t)n * 4 * H + j;
    o[0] = dc * gg * gi * (1.f - gi);
    o[H] = dc * cp * gf * (1.f - gf);
    o[2 * H] = dc * gi * (1.f - gg * gg);
    o[3 * H] = dh * tc * go * (1.f - go);
    d.dc[(int64_t)n * H + j] = dc * gf;
  }
}

int fill_args(StepArgs& a, const dvae_lstm_dir_t* dirs, int ndir, int T, int N, int H, int64_t ldh, bool bwd) {
  if (!dirs || ndir < 1 || ndir > 2 || T < 1 || N < 1 || H < 64 || (H & 63) || (ldh & 3)) return DVAE_EINVAL;
  for (int i = 0; i < ndir; ++i) {
    const dvae_lstm_dir_t& s = dirs[i];
    if (!s.gates || !s.w_hh || !s.c_all) return DVAE_EINVAL;
    if (bwd ? (!s.dh_out || !s.dgates || !s.dc_ws) : (!s.h_out)) return DVAE_EINVAL;
    if ((((uintptr_t)s.w_hh) | ((uintptr_t)s.h_out) | ((uintptr_t)s.dgates)) & 15) return DVAE_EINVAL;
    a.d[i].gates = s.gates; a.d[i].w = s.w_hh; a.d[i].h_out = s.h_out; a.d[i].c_all = s.c_all;
    a.d[i].dh_out = s.dh_out; a.d[i].dgates = s.dgates; a.d[i].dc = s.dc_ws; a.d[i].reverse = s.reverse;
  }
  if (ndir == 1) a.d[1] = a.d[0];
  a.T = T; a.N = N; a.H = H; a.ldh = ldh;
  return DVAE_OK;
}

// rows per workgroup: prefer the largest MT that still yields >= 256 workgroups
int pick_mt(int N, int H, int ndir) {
  const int n_j = H / 16;
  for (int mt = 2; mt >= 1; --mt) {
    const int n_m = (N + 16 * mt - 1) / (16 * mt);
    if (n_j * n_m * ndir >= 256 || mt == 1) return mt;
  }
  return 1;
}

}  // namespace

DVAE_API int dvae_lstm_seq_fwd(const dvae_lstm_dir_t* dirs, int ndir, int T, int N, int H, int64_t ldh,
                               void* stream) {
  StepArgs a{};
  int rc = fill_args(a, dirs, ndir, T, N, H, ldh, false);
  if (rc) return rc;
  hipStream_t s = (hipStream_t)stream;
  const int mt = pick_mt(N, H, ndir);
  const int n_j = H / 16, n_m = (N + 16 * mt - 1) / (16 * mt);
  dim3 grid(n_j * n_m, 1, ndir), block(256);
  ProfScope prof(2, s, 2.0 * N * 4.0 * H * H * (T - 1) * ndir);
  for (int step = 0; step < T; ++step) {
    if (mt == 2)
      hipLaunchKernelGGL((lstm_step_fwd_kernel<2>), grid, block, 0, s, a, step, n_j, n_m);
    else
      hipLaunchKernelGGL((lstm_step_fwd_kernel<1>), grid, block, 0, s, a, step, n_j, n_m);
  }
  return dvae_check_launch();
}

DVAE_API int dvae_lstm_seq_bwd(const dvae_lstm_dir_t* dirs, int ndir, int T, int N, int H, int64_t ldh,
                               void* stream) {
  StepArgs a{};
  int rc = fill_args(a, dirs, ndir, T, N, H, ldh, true);
  if (rc) return rc;
  hipStream_t s = (hipStream_t)stream;
  const int mt = pick_mt(N, H, ndir);
  const int n_j = H / 16, n_m = (N + 16 * mt - 1) / (16 * mt);
  dim3 grid(n_j * n_m, 1, ndir), block(256);
  ProfScope prof(2, s, 2.0 * N * 4.0 * H * H * (T - 1) * ndir);
  for (int step = 0; step < T; ++step) {
    if (mt == 2)
      hipLaunchKernelGGL((lstm_step_bwd_kernel<2>), grid, block, 0, s, a, step, n_j, n_m);
    else
      hipLaunchKernelGGL((lstm_step_bwd_kernel<1>), grid, block, 0, s, a, step, n_j, n_m);
  }
  return dvae_check_launch();
}
