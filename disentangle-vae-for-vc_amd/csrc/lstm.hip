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
//
// Data movement: per 64-deep k-chunk the workgroup stages its W_hh slice (64 rows) and its
// H[t-1] / dG[t+1] rows through LDS with full-line coalesced 16-byte loads (each byte fetched once
// per workgroup; the waves share the activation rows), register-prefetching chunk c+1 while the
// MFMAs of chunk c run, two LDS buffers, one barrier per chunk.  Fragments are read with
// ds_read_b128: lane (r, q) takes k = 4q..4q+3 of its row for both operands, which permutes the
// k order inside a 16-chunk identically for A and B and leaves the dot product intact.
// The operands of the pointwise epilogue (pre-activations, c[t-1], ...) are fetched at kernel
// entry so their latency hides under the contraction.  W_hh is re-read every frame and stays
// L2/MALL resident; the blockIdx -> tile map keeps all row-tiles of one weight slice on one XCD.
#include <cstdlib>
#include <type_traits>
#include "common.h"

// Streaming operands of a frame (pre-activations, cell state, h, dgates: touched once per frame) can be tagged
// non-temporal so that they do not push the re-used W_hh fragments out of the XCD's L2.
#ifndef DVAE_LSTM_NT
#define DVAE_LSTM_NT 0
#endif
#ifndef DVAE_LSTM_PF
#define DVAE_LSTM_PF 2     // rounds of operands in flight in the backward frame kernel (2 or 4)
#endif
#ifndef DVAE_LSTM_PFF
#define DVAE_LSTM_PFF 2    // ... and in the forward frame kernel
#endif
#if DVAE_LSTM_NT
#define LD_S(p) __builtin_nontemporal_load(p)
#define ST_S(p, v) __builtin_nontemporal_store((v), (p))
#else
#define LD_S(p) (*(p))
#define ST_S(p, v) (*(p) = (v))
#endif

namespace {

struct StepDir {
  float* gates;         // [T,N,4H]
  const float* w;       // fwd [4H,H] ; bwd W_hh^T [H,4H]
  const float* wp;      // fragment-packed weights for the v5 kernels (dvae_lstm_pack_w), or null
  float* h_out;         // [T,N,ldh] (column offset already applied)
  float* c_all;         // [T,N,H]
  const float* dh_out;  // [T,N,ldh]
  float* dgates;        // [T,N,4H]
  float* dc;            // [N,H]
  int reverse;
  int shift;            // this entry runs its (local) step s during the launch of global step s + shift (v5 kernels):
                        // two STACKED layers ride in one launch, layer 2 a chunk of frames behind layer 1
};
struct StepArgs {
  StepDir d[2];
  int T, N, H;
  int64_t ldh;
  int st16;  // bf16 mode: h_out (forward) / dgates (backward) are stored as bf16
  int pm;    // precision mode of the packed weights: 0 fp32 fragments, 1 bf16 (dvae_lstm_pack_w_bf16), 2 three bf16 planes (.._x3)
  int64_t ldg;   // row stride of gates / dgates (4H unless the H = 64 directions share one [T*N, 8H] tensor)
};

constexpr int KC = 64;   // k-chunk
constexpr int LDS_LD = 68;  // floats per staged row (64 + 4 pad; 272 B keeps 16-B alignment)

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


// Two-deep software pipeline of the k-chunks: chunk c is consumed from LDS while the global loads of
// chunks c+1 AND c+2 are in flight in two named register sets (so a load has two chunk-times, ~1 us,
// to come back from L2 before it is needed), two LDS buffers, one barrier per chunk.
// gload(w, a, c): issue loads of chunk c into the register set; sstore(buf, w, a): write the set to
// LDS buffer buf; compute(buf): MFMAs on LDS buffer buf.
template <int NRW, int NRA, class GL, class SS, class CP>
__device__ __forceinline__ void chunk_pipeline(int nchunks, GL gload, SS sstore, CP compute) {
  // Loads and LDS stores are UNCONDITIONAL (chunk index clamped to the last chunk, a few redundant loads
  // at the tail): with branch-free memory traffic hipcc keeps counted s_waitcnt vmcnt(N) in the loop, so
  // one register set really stays in flight across the barrier; conditional loads made it drain to
  // vmcnt(0) at the loop head.
  f32x4 wA[NRW], aA[NRA], wB[NRW], aB[NRA];
  const int last = nchunks - 1;
  gload(wA, aA, 0);
  gload(wB, aB, min(1, last));
  sstore(0, wA, aA);
  __syncthreads();
  for (int c = 0; c < nchunks; c += 2) {
    gload(wA, aA, min(c + 2, last));
    compute(0);
    sstore(1, wB, aB);
    __syncthreads();
    gload(wB, aB, min(c + 3, last));
    if (c + 1 < nchunks) compute(1);
    sstore(0, wA, aA);
    __syncthreads();
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

  __shared__ __attribute__((aligned(16))) float Ws[2][64 * LDS_LD];
  __shared__ __attribute__((aligned(16))) float As[2][16 * MT * LDS_LD];
  __shared__ float sm[4][MT * 16][17];

  // ---- epilogue operands first (independent of the contraction)
  float* __restrict__ G = d.gates + (int64_t)t * N * 4 * H;
  float pre[MT][4], cp[MT];
#pragma unroll
  for (int e = 0; e < MT; ++e) {
    const int idx = tid + 256 * e;
    const int n = m0 + (idx >> 4), j = j0 + (idx & 15);
    const bool ok = n < N;
#pragma unroll
    for (int g = 0; g < 4; ++g) pre[e][g] = ok ? G[(int64_t)n * 4 * H + g * H + j] : 0.f;
    cp[e] = (ok && step > 0) ? d.c_all[((int64_t)tp * N + n) * H + j] : 0.f;
  }

  f32x4 acc[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};

  if (step > 0) {
    const float* __restrict__ hp = d.h_out + (int64_t)tp * N * a.ldh;
    // staging map: 16 lanes cover one 256-B row segment
    const int srow = tid >> 4, sc4 = tid & 15;
    auto gload = [&](f32x4 (&rw)[4], f32x4 (&ra)[MT], int c) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = srow + 16 * i;  // 0..63 : gate = row>>4, j = row&15
        rw[i] = *reinterpret_cast<const f32x4*>(d.w + ((int64_t)(row >> 4) * H + j0 + (row & 15)) * H + c * KC + 4 * sc4);
      }
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const int n = min(m0 + srow + 16 * i, N - 1);   // rows >= N only feed output rows that are never stored
        ra[i] = *reinterpret_cast<const f32x4*>(hp + (int64_t)n * a.ldh + c * KC + 4 * sc4);
      }
    };
    auto sstore = [&](int buf, f32x4 (&rw)[4], f32x4 (&ra)[MT]) {
#pragma unroll
      for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(&Ws[buf][(srow + 16 * i) * LDS_LD + 4 * sc4]) = rw[i];
#pragma unroll
      for (int i = 0; i < MT; ++i) *reinterpret_cast<f32x4*>(&As[buf][(srow + 16 * i) * LDS_LD + 4 * sc4]) = ra[i];
    };
    auto compute = [&](int buf) {
      const float* __restrict__ wl = &Ws[buf][(wave * 16 + r) * LDS_LD + 4 * kq];
      const float* __restrict__ al = &As[buf][r * LDS_LD + 4 * kq];
      f32x4 bq[KC / 16], aq[MT][KC / 16];
#pragma unroll
      for (int kk = 0; kk < KC / 16; ++kk) {
        bq[kk] = *reinterpret_cast<const f32x4*>(wl + 16 * kk);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) aq[mt][kk] = *reinterpret_cast<const f32x4*>(al + mt * 16 * LDS_LD + 16 * kk);
      }
#pragma unroll
      for (int kk = 0; kk < KC / 16; ++kk)
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
            acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(aq[mt][kk][e], bq[kk][e], acc[mt], 0, 0, 0);
    };
    chunk_pipeline<4, MT>(H / KC, gload, sstore, compute);
  }
  // C/D map of the 16x16 tile: col = lane&15, row = (lane>>4)*4 + reg
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int e = 0; e < 4; ++e) sm[wave][mt * 16 + kq * 4 + e][r] = acc[mt][e];
  __syncthreads();

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
    const float gi = gate_sigmoid(sm[0][row][col] + pre[e][0]);
    const float gf = gate_sigmoid(sm[1][row][col] + pre[e][1]);
    const float gg = gate_tanh(sm[2][row][col] + pre[e][2]);
    const float go = gate_sigmoid(sm[3][row][col] + pre[e][3]);
    const float c = gf * cp[e] + gi * gg;
    g[0] = gi;
    g[H] = gf;
    g[2 * H] = gg;
    g[3 * H] = go;
    cout[(int64_t)n * H + j] = c;
    hout[(int64_t)n * a.ldh + j] = go * gate_tanh(c);
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

  __shared__ __attribute__((aligned(16))) float Bs[2][4 * 16 * LDS_LD];        // [k-quarter][j row]
  __shared__ __attribute__((aligned(16))) float As[2][4 * 16 * MT * LDS_LD];   // [k-quarter][segment row]
  __shared__ float sm[4][MT * 16][17];

  // ---- epilogue operands first
  const float* __restrict__ G = d.gates + (int64_t)t * N * 4 * H;
  float gt[MT][4], cc[MT], cp[MT], dho[MT], dcar[MT];
#pragma unroll
  for (int e = 0; e < MT; ++e) {
    const int idx = tid + 256 * e;
    const int n = m0 + (idx >> 4), j = j0 + (idx & 15);
    const bool ok = n < N;
#pragma unroll
    for (int g = 0; g < 4; ++g) gt[e][g] = ok ? G[(int64_t)n * 4 * H + g * H + j] : 0.f;
    cc[e] = ok ? d.c_all[((int64_t)t * N + n) * H + j] : 0.f;
    cp[e] = (ok && fstep > 0) ? d.c_all[((int64_t)tp * N + n) * H + j] : 0.f;
    dho[e] = ok ? d.dh_out[((int64_t)t * N + n) * a.ldh + j] : 0.f;
    dcar[e] = (ok && step > 0) ? d.dc[(int64_t)n * H + j] : 0.f;
  }

  f32x4 acc[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};

  if (step > 0) {
    // dHrec[n, j] = sum_k dG[tn][n, k] * W_hh[k, j] ; wave w: k in [w*H, (w+1)*H)
    const float* __restrict__ dgn = d.dgates + (int64_t)tn * N * 4 * H;
    const int H4 = 4 * H;
    const int srow = tid >> 4, sc4 = tid & 15;   // 16 rows x 16 float4 per pass
    auto gload = [&](f32x4 (&rb)[4], f32x4 (&ra)[4 * MT], int c) {
#pragma unroll
      for (int q = 0; q < 4; ++q)
        rb[q] = *reinterpret_cast<const f32x4*>(d.w + (int64_t)(j0 + srow) * H4 + q * H + c * KC + 4 * sc4);
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          const int n = min(m0 + srow + 16 * i, N - 1);
          ra[q * MT + i] = *reinterpret_cast<const f32x4*>(dgn + (int64_t)n * H4 + q * H + c * KC + 4 * sc4);
        }
    };
    auto sstore = [&](int buf, f32x4 (&rb)[4], f32x4 (&ra)[4 * MT]) {
#pragma unroll
      for (int q = 0; q < 4; ++q)
        *reinterpret_cast<f32x4*>(&Bs[buf][(q * 16 + srow) * LDS_LD + 4 * sc4]) = rb[q];
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int i = 0; i < MT; ++i)
          *reinterpret_cast<f32x4*>(&As[buf][(q * 16 * MT + srow + 16 * i) * LDS_LD + 4 * sc4]) = ra[q * MT + i];
    };
    auto compute = [&](int buf) {
      const float* __restrict__ bl = &Bs[buf][(wave * 16 + r) * LDS_LD + 4 * kq];
      const float* __restrict__ al = &As[buf][(wave * 16 * MT + r) * LDS_LD + 4 * kq];
      f32x4 bq[KC / 16], aq[MT][KC / 16];
#pragma unroll
      for (int kk = 0; kk < KC / 16; ++kk) {
        bq[kk] = *reinterpret_cast<const f32x4*>(bl + 16 * kk);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) aq[mt][kk] = *reinterpret_cast<const f32x4*>(al + mt * 16 * LDS_LD + 16 * kk);
      }
#pragma unroll
      for (int kk = 0; kk < KC / 16; ++kk)
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
            acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(aq[mt][kk][e], bq[kk][e], acc[mt], 0, 0, 0);
    };
    chunk_pipeline<4, 4 * MT>(H / KC, gload, sstore, compute);
  }
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int e = 0; e < 4; ++e) sm[wave][mt * 16 + kq * 4 + e][r] = acc[mt][e];
  __syncthreads();

  float* __restrict__ dG = d.dgates + (int64_t)t * N * 4 * H;
#pragma unroll
  for (int e = 0; e < MT; ++e) {
    const int idx = tid + 256 * e;
    const int row = idx >> 4, col = idx & 15;
    const int n = m0 + row;
    if (n >= N) continue;
    const int j = j0 + col;
    const float dh = dho[e] + sm[0][row][col] + sm[1][row][col] + sm[2][row][col] + sm[3][row][col];
    const float gi = gt[e][0], gf = gt[e][1], gg = gt[e][2], go = gt[e][3];
    const float tc = gate_tanh(cc[e]);
    const float dc = dcar[e] + dh * go * (1.f - tc * tc);
    float* o = dG + (int64_t)n * 4 * H + j;
    o[0] = dc * gg * gi * (1.f - gi);
    o[H] = dc * cp[e] * gf * (1.f - gf);
    o[2 * H] = dc * gi * (1.f - gg * gg);
    o[3 * H] = dh * tc * go * (1.f - go);
    d.dc[(int64_t)n * H + j] = dc * gf;
  }
}


// =====================================================================================================
// v5 frame kernels (H a multiple of 512): EIGHT waves per workgroup over a 16-unit x 16/32-segment tile.
// A W_hh row is consumed by exactly one wave, so staging it through LDS only costs ds_write bandwidth and barriers;
// but fragment-shaped global loads (16 rows x 64 B per instruction) are slow on the texture path.  So the weights are
// RE-PACKED into fragment order (dvae_lstm_pack_w): the 64 lanes of a wave read one contiguous 1-KiB burst per
// 16-deep k-chunk, straight into registers, a whole round ahead.  Only activation rows go through LDS: forward H[t-1]
// (shared by the four gate waves) in a few large block-level rounds; backward dG[t+1] in WAVE-PRIVATE staging (each
// wave owns a k-range), so the backward main loop has no workgroup barrier at all.  Loads are unconditional (row /
// round indices clamped) so hipcc keeps counted vmcnt waits.
// Measured on MI355X: a 32-row tile halves the W_hh re-read traffic but leaves one wave per SIMD
// (latency-bound, 21 us/frame); a 16-row tile gives two waves per SIMD but doubles the traffic (18 us).
// Eight waves = 4 gates (fwd) / k-quarters (bwd) x 2 k-halves give both: two waves per SIMD AND the
// small traffic; the k-halves meet in the LDS reduction that the epilogue needs anyway.
// =====================================================================================================
// PM (precision mode of the recurrent product):
//   0  fp32: W_hh fragments for v_mfma_f32_16x16x4_f32, fp32 h tile in LDS;
//   1  bf16 compute mode: W_hh packed as bf16 in the fragment order of v_mfma_f32_16x16x32_bf16 (lane (r, q) holds
//      k = 32c + 8q + j, j = 0..7, of gate column r: one 1-KiB burst per wave per chunk, 32 deep), the h tile rounded to
//      bf16 while it is staged into LDS (272-B rows: conflict-free ds_read_b128), fp32 accumulation and gates;
//   2  fp32x3: fp32 RESULTS on the bf16 pipe (see gemm.hip): W_hh packed as THREE bf16 planes (w = w1 + w2 + w3 exactly,
//      dvae_lstm_pack_w_x3), the h tile split the same way while it is staged (three LDS planes), six exact partial
//      products per (tile, chunk).  The fp32 MFMA is 16x slower than the bf16 one: at 70 % efficiency it was 9.7 us of a
//      14.5 us H = 1024 frame; the price is 1.5x the W_hh bytes.
// S16 (bf16 mode only): the state h is STORED as bf16 — the epilogue writes it so, the next frame stages it without a
// conversion, and the contractions that consume it (next layer's input projection, dW_hh) read 2 bytes per element.
template <int MT, int KR, int PM = 0, int PF = 2, bool S16 = false>
__global__ __launch_bounds__(512, (KR <= 64 && MT <= 2 ? 4 : 2)) void lstm_step_fwd_v5(const StepArgs a, int gstep, int n_j, int n_m) {
  static_assert(!S16 || PM == 1, "bf16 state storage: bf16 mode only");
  using st_t = typename std::conditional<S16, bf16x4, f32x4>::type;     // four staged h values
  using h_t = typename std::conditional<S16, __bf16, float>::type;
  constexpr int NW = 8;
  constexpr bool B16 = (PM != 0);
  constexpr int NP = (PM == 2) ? 3 : 1;            // bf16 planes per operand
  constexpr int KC = B16 ? 32 : 16;                // k depth of one packed chunk
  constexpr int NS = KR / KC, LDA = KR + (B16 ? 8 : 4);
  using lds_t = typename std::conditional<B16, __bf16, float>::type;
  constexpr int NST = MT * KR / 64;     // float4 per thread per round (activation stage, both k-halves together)
  const StepDir& d = a.d[blockIdx.z];
  const int step = gstep - d.shift;
  if (step < 0 || step >= a.T) return;        // this entry has no frame in this launch (uniform per workgroup)
  const int H = a.H, N = a.N;
  const int t = d.reverse ? (a.T - 1 - step) : step;
  const int tp = d.reverse ? t + 1 : t - 1;
  int jb, mb;
  decode_block(blockIdx.x, n_j, n_m, jb, mb);
  const int j0 = jb * 16, m0 = mb * 16 * MT;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int gate = wave & 3, kh = wave >> 2;
  const int r = lane & 15, kq = lane >> 4;

  // h tiles [buffer][k-half][plane][16*MT rows][LDA]; the gate tiles of the epilogue (sm) reuse the same bytes once the
  // last round has been computed (every round ends with a workgroup barrier), so two workgroups fit in a CU's 160 KB
  constexpr int PL = 16 * MT * LDA;                 // elements of one plane
  constexpr int AS_BYTES = 2 * 2 * NP * PL * (int)sizeof(lds_t), SM_BYTES = NW * MT * 16 * 17 * 4;
  __shared__ __attribute__((aligned(16))) char lds_raw[AS_BYTES > SM_BYTES ? AS_BYTES : SM_BYTES];
  lds_t (*As)[2][NP * PL] = reinterpret_cast<lds_t (*)[2][NP * PL]>(lds_raw);
  float (*sm)[MT * 16][17] = reinterpret_cast<float (*)[MT * 16][17]>(lds_raw);

  // epilogue operands: (segment, unit) elements of this thread (MT*256 of them over 512 threads)
  constexpr int NE = (MT * 256 + 511) / 512;
  float* __restrict__ G = d.gates + (int64_t)t * N * 4 * H;
  int en[NE], ej[NE];
  bool eok[NE];
  float pre[NE][4], cp[NE];
#pragma unroll
  for (int e = 0; e < NE; ++e) {
    const int idx = tid + 512 * e;
    en[e] = m0 + (idx >> 4);
    ej[e] = j0 + (idx & 15);
    eok[e] = (idx < MT * 256) && (en[e] < N);
#pragma unroll
    for (int g = 0; g < 4; ++g) pre[e][g] = eok[e] ? LD_S(&G[(int64_t)en[e] * 4 * H + g * H + ej[e]]) : 0.f;
    cp[e] = (eok[e] && step > 0) ? LD_S(&d.c_all[((int64_t)tp * N + en[e]) * H + ej[e]]) : 0.f;
  }

  f32x4 acc[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};

  if (step > 0) {
    const h_t* __restrict__ hp = reinterpret_cast<const h_t*>(d.h_out) + (int64_t)tp * N * a.ldh;
    const int nr = H / KR / 2, last = nr - 1;           // rounds per k-half
    // packed W: [(gate*n_j + jb)][k-chunk][plane][lane][4 dwords]; this wave's chunks start at kh*(H/KC/2)
    const float* __restrict__ wpk = d.wp + (((int64_t)gate * n_j + jb) * (H / KC) + (int64_t)kh * (H / KC / 2)) * (256 * NP) + lane * 4;
    // staging: threads 0..255 stage k-half 0, 256..511 k-half 1; 16 lanes per 256-B row segment
    const int skh = tid >> 8, srow = (tid & 255) >> 4, sc4 = tid & 15;
    auto loadW = [&](f32x4 (&w)[NS * NP], int rd) {
#pragma unroll
      for (int s = 0; s < NS * NP; ++s) w[s] = *reinterpret_cast<const f32x4*>(wpk + (int64_t)(rd * NS * NP + s) * 256);
    };
    auto loadA = [&](st_t (&st)[NST], int rd) {
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const int n = min(m0 + srow + 16 * i, N - 1);
#pragma unroll
        for (int q = 0; q < KR / 64; ++q)
          st[i * (KR / 64) + q] = *reinterpret_cast<const st_t*>(hp + (int64_t)n * a.ldh + (skh * nr + rd) * KR + 64 * q + 4 * sc4);
      }
    };
    auto storeA = [&](int buf, st_t (&st)[NST]) {
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int q = 0; q < KR / 64; ++q) {
          const int off = (srow + 16 * i) * LDA + 64 * q + 4 * sc4;
          const st_t v = st[i * (KR / 64) + q];
          if constexpr (S16) {
            *reinterpret_cast<bf16x4*>(&As[buf][skh][off]) = v;
          } else if constexpr (PM == 2) {
            bf16x4 pl[3];
            split3(v, pl);
#pragma unroll
            for (int p = 0; p < 3; ++p) *reinterpret_cast<bf16x4*>(&As[buf][skh][p * PL + off]) = pl[p];
          } else if constexpr (PM == 1) {
            *reinterpret_cast<bf16x4*>(&As[buf][skh][off]) = __builtin_convertvector(v, bf16x4);
          } else {
            *reinterpret_cast<f32x4*>(&As[buf][skh][off]) = v;
          }
        }
    };
    auto compute = [&](int buf, f32x4 (&w)[NS * NP]) {
      if constexpr (B16) {
        const lds_t* __restrict__ al = &As[buf][kh][r * LDA + 8 * kq];
        // fp32x3: the six partial products of weight >= 2^-16 (h plane, W plane)
        constexpr int NT6 = (PM == 2) ? 6 : 1;
        constexpr int ia[6] = {0, 0, 1, 1, 0, 2}, ib[6] = {0, 1, 0, 1, 2, 0};
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
            bf16x8 av[NP];
#pragma unroll
            for (int p = 0; p < NP; ++p) av[p] = *reinterpret_cast<const bf16x8*>(al + p * PL + mt * 16 * LDA + 32 * s);
#pragma unroll
            for (int term = 0; term < NT6; ++term)
              acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[ia[term]], __builtin_bit_cast(bf16x8, w[s * NP + ib[term]]),
                                                                acc[mt], 0, 0, 0);
          }
      } else {
      const lds_t* __restrict__ al = &As[buf][kh][r * LDA + 4 * kq];
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        f32x4 av[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) av[mt] = *reinterpret_cast<const f32x4*>(al + mt * 16 * LDA + 16 * s);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
            acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mt][e], w[s][e], acc[mt], 0, 0, 0);
      }
      }
    };
    // PF register sets (W fragments + h rows of PF rounds) in flight, see the backward kernel; PF divides nr
    f32x4 wS[PF][NS * NP];
    st_t sS[PF][NST];
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      loadA(sS[u], min(u, last));
      loadW(wS[u], min(u, last));
    }
    storeA(0, sS[0]);
    __syncthreads();
    for (int rd0 = 0; rd0 < nr; rd0 += PF) {
#pragma unroll
      for (int u = 0; u < PF; ++u) {
        __builtin_amdgcn_sched_barrier(0);   // keep the prefetch loads where they are issued (hipcc otherwise sinks them)
        compute(u & 1, wS[u]);
        __builtin_amdgcn_sched_barrier(0);
        storeA((u + 1) & 1, sS[(u + 1) % PF]);
        __syncthreads();
        loadA(sS[u], min(rd0 + u + PF, last));
        loadW(wS[u], min(rd0 + u + PF, last));
      }
    }
  }
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int e = 0; e < 4; ++e) sm[wave][mt * 16 + kq * 4 + e][r] = acc[mt][e];
  __syncthreads();

#pragma unroll
  for (int e = 0; e < NE; ++e) {
    if (!eok[e]) continue;
    const int idx = tid + 512 * e;
    const int row = idx >> 4, col = idx & 15;
    float* g = G + (int64_t)en[e] * 4 * H + ej[e];
    const float gi = gate_sigmoid(sm[0][row][col] + sm[4][row][col] + pre[e][0]);
    const float gf = gate_sigmoid(sm[1][row][col] + sm[5][row][col] + pre[e][1]);
    const float gg = gate_tanh(sm[2][row][col] + sm[6][row][col] + pre[e][2]);
    const float go = gate_sigmoid(sm[3][row][col] + sm[7][row][col] + pre[e][3]);
    const float c = gf * cp[e] + gi * gg;
    ST_S(&g[0], gi);
    ST_S(&g[H], gf);
    ST_S(&g[2 * H], gg);
    ST_S(&g[3 * H], go);
    ST_S(&d.c_all[((int64_t)t * N + en[e]) * H + ej[e]], c);
    reinterpret_cast<h_t*>(d.h_out)[((int64_t)t * N + en[e]) * a.ldh + ej[e]] = (h_t)(go * gate_tanh(c));
  }
}

// (An "all loads of the frame in flight up front" variant -- 217 VGPRs, no spills -- measured 17.1 us/frame against
// 15.9 us for the just-in-time prefetch above: flooding the L2 queues delays the first tile of every workgroup.)

// PM as in the forward kernel.  KR = k depth of a round (64; 32 in fp32x3 mode, where the wave-private staging holds three
// planes per buffer: 8 waves x 2 buffers x 3 planes x 32 rows x 40 bf16 = 123 KB).
// NU = 16-unit column tiles per workgroup.  NU = 2 (32 hidden units x 16*MT segments) halves the dG[t+1] stream (4H wide,
// the dominant traffic of this kernel) and turns two stacked H = 1024 layers into ONE resident round of 256 workgroups
// instead of two — and measured 24.8 us per layer-frame against 18.1 (each wave's serial chain doubles; nothing overlaps
// it at one workgroup per CU).  Kept as a parameter, dispatched with NU = 1.
// S16 (bf16 mode only): the gate gradients dG are STORED as bf16 (written so by the epilogue, staged without conversion,
// read as bf16 by the dx / dW contractions and the bias column sums)
template <int MT, int KR, int PM = 0, int NU = 1, bool S16 = false>
__global__ __launch_bounds__(512) void lstm_step_bwd_v5(const StepArgs a, int gstep, int n_j, int n_m) {
  static_assert(!S16 || PM == 1, "bf16 gate-gradient storage: bf16 mode only");
  using st_t = typename std::conditional<S16, bf16x4, f32x4>::type;
  using g_t = typename std::conditional<S16, __bf16, float>::type;
  constexpr int NW = 8;
  constexpr bool B16 = (PM != 0);
  constexpr int NP = (PM == 2) ? 3 : 1;
  constexpr int KC = B16 ? 32 : 16;
  constexpr int NS = KR / KC, LDA = KR + (B16 ? 8 : 4);
  constexpr int PL = 16 * MT * LDA;                       // elements of one plane of one buffer
  constexpr int KRW = KR >= 64 ? 64 : KR;                 // floats of a row one load instruction covers
  constexpr int LPR = KRW / 4, RPI = 64 / LPR;            // lanes per row, rows per instruction
  constexpr int NRI = 16 * MT / RPI, NQ = KR / KRW;       // row groups, instructions per row
  constexpr int UW = 16 * NU;                             // hidden units per workgroup
  using lds_t = typename std::conditional<B16, __bf16, float>::type;
  const StepDir& d = a.d[blockIdx.z];
  const int step = gstep - d.shift;
  if (step < 0 || step >= a.T) return;
  const int H = a.H, N = a.N;
  const int fstep = a.T - 1 - step;
  const int t = d.reverse ? (a.T - 1 - fstep) : fstep;
  const int tn = d.reverse ? t - 1 : t + 1;
  const int tp = d.reverse ? t + 1 : t - 1;
  int jb, mb;
  decode_block(blockIdx.x, n_j, n_m, jb, mb);
  const int j0 = jb * UW, m0 = mb * 16 * MT;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int quarter = wave >> 1, part = wave & 1;      // k range: [quarter*H + part*H/2, +H/2)
  const int r = lane & 15, kq = lane >> 4;

  // per wave: two staging buffers of NP planes; once a wave has left its k loop the same bytes hold its partial tiles
  // (sm) for the cross-wave reduction (wave-private until the workgroup barrier)
  constexpr int ST_BYTES = 2 * NP * PL * (int)sizeof(lds_t), SM_BYTES = MT * 16 * (UW + 1) * 4;
  constexpr int WAVE_BYTES = ((ST_BYTES > SM_BYTES ? ST_BYTES : SM_BYTES) + 15) / 16 * 16;
  __shared__ __attribute__((aligned(16))) char lds_raw[NW * WAVE_BYTES];
  lds_t* __restrict__ stg = reinterpret_cast<lds_t*>(lds_raw + wave * WAVE_BYTES);
  auto sm = [&](int w, int row, int col) -> float& {
    return reinterpret_cast<float*>(lds_raw + w * WAVE_BYTES)[row * (UW + 1) + col];
  };

  // epilogue operands: MT*16 x UW (segment, unit) elements over 512 threads
  constexpr int NEL = MT * 16 * UW, NE = (NEL + 511) / 512;
  const float* __restrict__ G = d.gates + (int64_t)t * N * 4 * H;
  int en[NE], ej[NE];
  bool eok[NE];
  float gt[NE][4], cc[NE], cp[NE], dho[NE], dcar[NE];
#pragma unroll
  for (int e = 0; e < NE; ++e) {
    const int idx = tid + 512 * e;
    en[e] = m0 + idx / UW;
    ej[e] = j0 + idx % UW;
    eok[e] = (idx < NEL) && (en[e] < N);
#pragma unroll
    for (int g = 0; g < 4; ++g) gt[e][g] = eok[e] ? G[(int64_t)en[e] * 4 * H + g * H + ej[e]] : 0.f;
    cc[e] = eok[e] ? d.c_all[((int64_t)t * N + en[e]) * H + ej[e]] : 0.f;
    cp[e] = (eok[e] && fstep > 0) ? d.c_all[((int64_t)tp * N + en[e]) * H + ej[e]] : 0.f;
    dho[e] = eok[e] ? d.dh_out[((int64_t)t * N + en[e]) * a.ldh + ej[e]] : 0.f;
    dcar[e] = (eok[e] && step > 0) ? d.dc[(int64_t)en[e] * H + ej[e]] : 0.f;
  }

  f32x4 acc[NU][MT];
#pragma unroll
  for (int u = 0; u < NU; ++u)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[u][mt] = f32x4{0.f, 0.f, 0.f, 0.f};

  if (step > 0) {
    const int H4 = 4 * H;
    const int koff = quarter * H + part * (H / 2);
    // packed W^T: [(jb16*4 + quarter)][k-chunk inside the quarter][plane][lane][4 dwords], jb16 = 16-unit column block
    const float* __restrict__ bpk = d.wp + (((int64_t)jb * NU * 4 + quarter) * (H / KC) + (int64_t)part * (H / KC / 2)) * (256 * NP) + lane * 4;
    const int64_t bpk_u = (int64_t)4 * (H / KC) * (256 * NP);      // from one 16-unit block to the next
    const int lrow = lane / LPR, lc4 = lane % LPR;
    const g_t* arow[NRI];
#pragma unroll
    for (int i = 0; i < NRI; ++i) {
      const int n = min(m0 + lrow + RPI * i, N - 1);
      arow[i] = reinterpret_cast<const g_t*>(d.dgates) + ((int64_t)tn * N + n) * H4 + koff + 4 * lc4;
    }
    const int nr = H / 2 / KR, last = nr - 1;
    auto loadA = [&](st_t (&st)[NRI][NQ], int rd) {
#pragma unroll
      for (int i = 0; i < NRI; ++i)
#pragma unroll
        for (int q = 0; q < NQ; ++q) st[i][q] = *reinterpret_cast<const st_t*>(arow[i] + rd * KR + KRW * q);
    };
    auto storeA = [&](int buf, st_t (&st)[NRI][NQ]) {
#pragma unroll
      for (int i = 0; i < NRI; ++i)
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
          lds_t* dst = stg + buf * (NP * PL) + (lrow + RPI * i) * LDA + KRW * q + 4 * lc4;
          if constexpr (S16) {
            *reinterpret_cast<bf16x4*>(dst) = st[i][q];
          } else if constexpr (PM == 2) {
            bf16x4 pl[3];
            split3(st[i][q], pl);
#pragma unroll
            for (int p = 0; p < 3; ++p) *reinterpret_cast<bf16x4*>(dst + p * PL) = pl[p];
          } else if constexpr (PM == 1) {
            *reinterpret_cast<bf16x4*>(dst) = __builtin_convertvector(st[i][q], bf16x4);
          } else {
            *reinterpret_cast<f32x4*>(dst) = st[i][q];
          }
        }
    };
    auto loadB = [&](f32x4 (&b)[NU][NS * NP], int rd) {
#pragma unroll
      for (int u = 0; u < NU; ++u)
#pragma unroll
        for (int s = 0; s < NS * NP; ++s)
          b[u][s] = *reinterpret_cast<const f32x4*>(bpk + u * bpk_u + (int64_t)(rd * NS * NP + s) * 256);
    };
    auto compute = [&](int buf, f32x4 (&b)[NU][NS * NP]) {
      if constexpr (B16) {
        const lds_t* __restrict__ al = stg + buf * (NP * PL) + r * LDA + 8 * kq;
        constexpr int NT6 = (PM == 2) ? 6 : 1;
        constexpr int ia[6] = {0, 0, 1, 1, 0, 2}, ib[6] = {0, 1, 0, 1, 2, 0};
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
            bf16x8 av[NP];
#pragma unroll
            for (int p = 0; p < NP; ++p) av[p] = *reinterpret_cast<const bf16x8*>(al + p * PL + mt * 16 * LDA + 32 * s);
#pragma unroll
            for (int term = 0; term < NT6; ++term)
#pragma unroll
              for (int u = 0; u < NU; ++u)
                acc[u][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[ia[term]], __builtin_bit_cast(bf16x8, b[u][s * NP + ib[term]]),
                                                                     acc[u][mt], 0, 0, 0);
          }
      } else {
      const lds_t* __restrict__ al = stg + buf * (NP * PL) + r * LDA + 4 * kq;
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        f32x4 av[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) av[mt] = *reinterpret_cast<const f32x4*>(al + mt * 16 * LDA + 16 * s);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int u = 0; u < NU; ++u)
              acc[u][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mt][e], b[u][s][e], acc[u][mt], 0, 0, 0);
      }
      }
    };
    // PF register sets (dG rows + W fragments of PF rounds) in flight (ring; loads unconditional, round index clamped).
    // Measured at H = 1024: PF = 4 is SLOWER than 2 (20.4 vs 18.0 us per frame: more loads in flight only deepen the
    // queues), so the frame kernels are throughput-, not latency-bound; see DESIGN.md.
    constexpr int PF = DVAE_LSTM_PF;
    f32x4 bS[PF][NU][NS * NP];
    st_t sS[PF][NRI][NQ];
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      loadA(sS[u], min(u, last));
      loadB(bS[u], min(u, last));
    }
    storeA(0, sS[0]);
    __builtin_amdgcn_wave_barrier();
    for (int rd0 = 0; rd0 < nr; rd0 += PF) {   // nr is a multiple of 4 (H a multiple of 512), PF divides it
#pragma unroll
      for (int u = 0; u < PF; ++u) {
        __builtin_amdgcn_sched_barrier(0);
        compute(u & 1, bS[u]);
        __builtin_amdgcn_sched_barrier(0);
        storeA((u + 1) & 1, sS[(u + 1) % PF]);     // the next round's rows, fetched PF-1 rounds ago
        __builtin_amdgcn_wave_barrier();
        loadA(sS[u], min(rd0 + u + PF, last));     // set u is free again
        loadB(bS[u], min(rd0 + u + PF, last));
      }
    }
  }
  __builtin_amdgcn_wave_barrier();     // this wave's staging reads are done: its bytes become its partial tiles
#pragma unroll
  for (int u = 0; u < NU; ++u)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int e = 0; e < 4; ++e) sm(wave, mt * 16 + kq * 4 + e, u * 16 + r) = acc[u][mt][e];
  __syncthreads();

#pragma unroll
  for (int e = 0; e < NE; ++e) {
    if (!eok[e]) continue;
    const int idx = tid + 512 * e;
    const int row = idx / UW, col = idx % UW;
    float rec = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) rec += sm(w, row, col);
    const float dh = dho[e] + rec;
    const float gi = gt[e][0], gf = gt[e][1], gg = gt[e][2], go = gt[e][3];
    const float tc = gate_tanh(cc[e]);
    const float dc = dcar[e] + dh * go * (1.f - tc * tc);
    g_t* o = reinterpret_cast<g_t*>(d.dgates) + ((int64_t)t * N + en[e]) * 4 * H + ej[e];
    o[0] = (g_t)(dc * gg * gi * (1.f - gi));
    o[H] = (g_t)(dc * cp[e] * gf * (1.f - gf));
    o[2 * H] = (g_t)(dc * gi * (1.f - gg * gg));
    o[3 * H] = (g_t)(dh * tc * go * (1.f - go));
    d.dc[(int64_t)en[e] * H + ej[e]] = dc * gf;
  }
}


// =====================================================================================================
// H = 64 (the encoder BiLSTM): the recurrence of a mel segment depends only on that segment's own rows, and at
// H = 64 one workgroup can hold W_hh in registers and do a whole frame's gate GEMM in ~1 us.  So a workgroup
// owns 16 segments of one direction and walks ALL T frames inside ONE launch -- no inter-workgroup dependency,
// no grid barrier, 2 launches per layer instead of 2*T.  (At H >= 512 a row-split would stream all of W_hh per
// workgroup per frame, so those sizes keep the one-launch-per-frame kernels above.)
// =====================================================================================================
// PM = 1 (bf16 compute mode): the recurrent product on v_mfma_f32_16x16x32_bf16 — W_hh rounded to bf16 once (RNE, in
// registers), h[t-1] rounded as it is written to LDS; everything else (gates, cell state, the stored h) stays fp32.
// PM = 2 (fp32x3, the default arithmetic): both operands split exactly into three bf16 planes (W_hh once, h[t-1] as it
// is written), six partial products per product: fp32 results.  On the fp32 MFMA (PM = 0) the 32 MFMAs of a wave are
// 2 048 of a frame's ~3 500 cycles per SIMD; with PM = 1 they are 4 of 16 cycles, with PM = 2 24.
__device__ __forceinline__ void split3s(float x, __bf16 (&p)[3]) {
  p[0] = (__bf16)x;
  float r = x - (float)p[0];
  p[1] = (__bf16)r;
  r -= (float)p[1];
  p[2] = (__bf16)r;       // exact: what is left has at most 8 significant bits
}
// Waves: PM = 0: 8 (gate, n-half: two 16-unit n-tiles each); PM != 0: 16 (gate, n-tile) and ONE element per thread in
// the gate arithmetic — with the MFMAs cheap, a frame is a chain of latencies (fragment reads, MFMAs, the gs round
// trip, ten transcendentals per element, two barriers), and sixteen waves halve the per-wave serial parts of it.
// ROWS (PM != 0): segments per workgroup.  The MFMA tile has 16 rows whatever ROWS is (24 MFMAs per (gate, n-tile) pair and
// frame in the split arithmetic: 768 cycles per SIMD), but the gate arithmetic — ten transcendentals per element — is per
// row: with N = 128 segments, 8 (4) rows per workgroup spread it over 32 (64) CUs instead of 16.
template <int PM, int ROWS = 16>
__global__ __launch_bounds__(PM ? 64 * ROWS : 512) void lstm_seq_fwd_h64(const StepArgs a) {
  constexpr int H = 64;
  constexpr bool B16 = PM != 0;
  constexpr int NP = PM == 2 ? 3 : 1;
  constexpr int NTHR = PM ? 64 * ROWS : 512, NE = ROWS * 64 / NTHR, NT = 16 / (NTHR / 64);
  static_assert(PM != 0 || ROWS == 16, "the fp32-MFMA form keeps 16 rows");
  const StepDir& d = a.d[blockIdx.y];
  const int N = a.N, T = a.T;
  const int m0 = blockIdx.x * ROWS;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int gate = wave & 3, part = wave >> 2;        // n-tiles part * NT .. + NT - 1
  const int r = lane & 15, kq = lane >> 4;

  __shared__ __attribute__((aligned(16))) float hs[B16 ? 1 : 16][68];     // h[t-1] rows, k-contiguous
  __shared__ __attribute__((aligned(16))) __bf16 hs16[NP][B16 ? 16 : 1][72];  // ... as bf16 plane(s) (144-byte rows: conflict-free b128 reads)
  __shared__ float gs[4][16][65];                                // recurrent pre-activation [gate][row][unit]

  f32x4 wf[NT][4];       // PM = 0: this wave's W_hh fragments: n-tiles x 4 k-chunks, resident for the whole sequence
  bf16x8 wb[NP][NT][2];  // B16: plane x n-tile x 2 k-chunks of 32 (lane (r, kq) holds k = 32 kc + 8 kq .. + 7 of unit row r)
#pragma unroll
  for (int ntl = 0; ntl < NT; ++ntl) {
    const float* wrow = d.w + ((int64_t)(gate * H + (part * NT + ntl) * 16 + r)) * H;
    if constexpr (B16) {
#pragma unroll
      for (int kc = 0; kc < 2; ++kc) {
        const f32x4 lo = *reinterpret_cast<const f32x4*>(wrow + kc * 32 + 8 * kq);
        const f32x4 hi = *reinterpret_cast<const f32x4*>(wrow + kc * 32 + 8 * kq + 4);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float wv = e < 4 ? lo[e & 3] : hi[e & 3];
          if constexpr (PM == 2) {
            __bf16 pl[3];
            split3s(wv, pl);
#pragma unroll
            for (int q = 0; q < 3; ++q) wb[q][ntl][kc][e] = pl[q];
          } else {
            wb[0][ntl][kc][e] = (__bf16)wv;
          }
        }
      }
    } else {
#pragma unroll
      for (int kc = 0; kc < 4; ++kc) wf[ntl][kc] = *reinterpret_cast<const f32x4*>(wrow + kc * 16 + 4 * kq);
    }
  }

  int erow[NE], ej[NE];
  bool eok[NE];
  float creg[NE];
#pragma unroll
  for (int e = 0; e < NE; ++e) {
    const int idx = tid + NTHR * e;
    erow[e] = idx >> 6;
    ej[e] = idx & 63;
    eok[e] = (m0 + erow[e]) < N;
    creg[e] = 0.f;
    if constexpr (B16) {
#pragma unroll
      for (int q = 0; q < NP; ++q) hs16[q][erow[e]][ej[e]] = (__bf16)0.f;
    } else {
      hs[erow[e]][ej[e]] = 0.f;
    }
  }
  if constexpr (B16 && ROWS < 16) {      // the tile's unused rows: zero for the whole sequence
    for (int idx = tid + ROWS * 64; idx < 16 * 64; idx += NTHR)
#pragma unroll
      for (int q = 0; q < NP; ++q) hs16[q][idx >> 6][idx & 63] = (__bf16)0.f;
  }
  __syncthreads();

  // The pre-activations of frame step+1 are fetched while frame step computes (their latency, ~1 us from L2, is
  // several times one frame's work here).  Two register sets that swap roles every frame (loop unrolled by two) and
  // branch-free loads (rows past N read row N-1 and are never stored): with `x = ok ? load : 0` + a copy per frame,
  // hipcc put every load in its own basic block and drained vmcnt(0) in the middle of the SAME frame's MFMAs.
  int nrow[NE];
#pragma unroll
  for (int e = 0; e < NE; ++e) nrow[e] = min(m0 + erow[e], N - 1);
  auto fetch = [&](int step_, float (&x)[NE][4]) {
    const int t_ = d.reverse ? (T - 1 - step_) : step_;
    const float* __restrict__ G_ = d.gates + (int64_t)t_ * N * a.ldg;
#pragma unroll
    for (int e = 0; e < NE; ++e)
#pragma unroll
      for (int g = 0; g < 4; ++g) x[e][g] = G_[(int64_t)nrow[e] * a.ldg + g * H + ej[e]];
  };
  auto frame = [&](int step, float (&xp)[NE][4], float (&xn)[NE][4]) {
    const int t = d.reverse ? (T - 1 - step) : step;
    float* __restrict__ G = d.gates + (int64_t)t * N * a.ldg;
    fetch(min(step + 1, T - 1), xn);

    f32x4 acc[NT];
#pragma unroll
    for (int ntl = 0; ntl < NT; ++ntl) acc[ntl] = f32x4{0.f, 0.f, 0.f, 0.f};
    if constexpr (B16) {
#pragma unroll
      for (int kc = 0; kc < 2; ++kc) {
        bf16x8 av[NP];
#pragma unroll
        for (int q = 0; q < NP; ++q) av[q] = *reinterpret_cast<const bf16x8*>(&hs16[q][r][kc * 32 + 8 * kq]);
        // partial products, small ones first: (2,0) (1,1) (0,2) (1,0) (0,1) (0,0)
        constexpr int ia[6] = {2, 1, 0, 1, 0, 0}, ib[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
        for (int term = (NP == 3 ? 0 : 5); term < 6; ++term)
#pragma unroll
          for (int ntl = 0; ntl < NT; ++ntl)
            acc[ntl] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[ia[term]], wb[ib[term]][ntl][kc], acc[ntl], 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int kc = 0; kc < 4; ++kc) {
        const f32x4 av = *reinterpret_cast<const f32x4*>(&hs[r][kc * 16 + 4 * kq]);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int ntl = 0; ntl < NT; ++ntl)
            acc[ntl] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[e], wf[ntl][kc][e], acc[ntl], 0, 0, 0);
      }
    }
#pragma unroll
    for (int ntl = 0; ntl < NT; ++ntl)
#pragma unroll
      for (int q = 0; q < 4; ++q) gs[gate][kq * 4 + q][(part * NT + ntl) * 16 + r] = acc[ntl][q];
    __syncthreads();
#pragma unroll
    for (int e = 0; e < NE; ++e) {
      const int row = erow[e], j = ej[e];
      const float gi = gate_sigmoid(gs[0][row][j] + xp[e][0]);
      const float gf = gate_sigmoid(gs[1][row][j] + xp[e][1]);
      const float gg = gate_tanh(gs[2][row][j] + xp[e][2]);
      const float go = gate_sigmoid(gs[3][row][j] + xp[e][3]);
      const float c = gf * creg[e] + gi * gg;
      const float h = go * gate_tanh(c);
      creg[e] = c;
      if constexpr (PM == 2) {
        __bf16 pl[3];
        split3s(eok[e] ? h : 0.f, pl);
#pragma unroll
        for (int q = 0; q < 3; ++q) hs16[q][row][j] = pl[q];
      } else if constexpr (PM == 1) {
        hs16[0][row][j] = (__bf16)(eok[e] ? h : 0.f);
      } else {
        hs[row][j] = eok[e] ? h : 0.f;
      }
      if (eok[e]) {
        const int64_t n = m0 + row;
        float* g = G + n * a.ldg + j;
        g[0] = gi;
        g[H] = gf;
        g[2 * H] = gg;
        g[3 * H] = go;
        d.c_all[((int64_t)t * N + n) * H + j] = c;
        d.h_out[((int64_t)t * N + n) * a.ldh + j] = h;
      }
    }
    __syncthreads();
  };
  float xa[NE][4], xb[NE][4];
  fetch(0, xa);
  for (int step = 0; step < T; step += 2) {
    frame(step, xa, xb);
    if (step + 1 < T) frame(step + 1, xb, xa);
  }
}

// PM = 1 / 2: dG[t+1] and W_hh as one / three bf16 planes for the recurrent product (see lstm_seq_fwd_h64).  Waves: (n-tile of
// 16 hidden units, k-part of the 256 gate columns): PM = 0: 8 waves, k-halves; PM != 0: 16 waves, k-quarters, one element
// per thread in the gate arithmetic.
// ROWS: segments per workgroup (PM != 0; see lstm_seq_fwd_h64): 4 ROWS / 16 waves = k-parts of the 256 gate columns.
template <int PM, int ROWS = 16>
__global__ __launch_bounds__(PM ? 64 * ROWS : 512) void lstm_seq_bwd_h64(const StepArgs a) {
  constexpr int H = 64;
  constexpr bool B16 = PM != 0;
  constexpr int NP = PM == 2 ? 3 : 1;
  constexpr int NTHR = PM ? 64 * ROWS : 512, NE = ROWS * 64 / NTHR, KP = NTHR / 256, KW = 256 / KP;   // k-parts, their width
  static_assert(PM != 0 || ROWS == 16, "the fp32-MFMA form keeps 16 rows");
  const StepDir& d = a.d[blockIdx.y];   // d.w = W_hh^T [H][4H]
  const int N = a.N, T = a.T;
  const int m0 = blockIdx.x * ROWS;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nt = wave & 3, kp = wave >> 2;
  const int r = lane & 15, kq = lane >> 4;

  __shared__ __attribute__((aligned(16))) float dgs[B16 ? 1 : 16][260];     // dG[t+1] rows (k = gate*64 + unit)
  __shared__ __attribute__((aligned(16))) __bf16 dgs16[NP][B16 ? 16 : 1][264];  // ... as bf16 plane(s) (528-byte rows)
  __shared__ float rs[KP][16][65];                              // partial dHrec per k-part

  f32x4 wf[8];
  bf16x8 wb[NP][KW / 32];     // B16: plane x k-chunks of 32 of this wave's k-part
  {
    const float* wrow = d.w + (int64_t)(nt * 16 + r) * 4 * H + kp * KW;
    if constexpr (B16) {
#pragma unroll
      for (int kc = 0; kc < KW / 32; ++kc) {
        const f32x4 lo = *reinterpret_cast<const f32x4*>(wrow + kc * 32 + 8 * kq);
        const f32x4 hi = *reinterpret_cast<const f32x4*>(wrow + kc * 32 + 8 * kq + 4);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float wv = e < 4 ? lo[e & 3] : hi[e & 3];
          if constexpr (PM == 2) {
            __bf16 pl[3];
            split3s(wv, pl);
#pragma unroll
            for (int q = 0; q < 3; ++q) wb[q][kc][e] = pl[q];
          } else {
            wb[0][kc][e] = (__bf16)wv;
          }
        }
      }
    } else {
#pragma unroll
      for (int kc = 0; kc < 8; ++kc) wf[kc] = *reinterpret_cast<const f32x4*>(wrow + kc * 16 + 4 * kq);
    }
  }

  int erow[NE], ej[NE];
  bool eok[NE];
  float dcreg[NE];
#pragma unroll
  for (int e = 0; e < NE; ++e) {
    const int idx = tid + NTHR * e;
    erow[e] = idx >> 6;
    ej[e] = idx & 63;
    eok[e] = (m0 + erow[e]) < N;
    dcreg[e] = 0.f;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      if constexpr (B16) {
#pragma unroll
        for (int q = 0; q < NP; ++q) dgs16[q][erow[e]][g * H + ej[e]] = (__bf16)0.f;
      } else {
        dgs[erow[e]][g * H + ej[e]] = 0.f;
      }
    }
  }
  if constexpr (B16 && ROWS < 16) {      // the tile's unused rows: zero for the whole sequence
    for (int idx = tid + ROWS * 64; idx < 16 * 64; idx += NTHR)
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int q = 0; q < NP; ++q) dgs16[q][idx >> 6][g * H + (idx & 63)] = (__bf16)0.f;
  }
  __syncthreads();

  // frame operands of step+1 in flight under frame step: two register sets swapping roles, branch-free clamped loads
  // (see lstm_seq_fwd_h64)
  struct Ops {
    float gt[NE][4], cc[NE], cp[NE], dho[NE];
  };
  int nrow[NE];
#pragma unroll
  for (int e = 0; e < NE; ++e) nrow[e] = min(m0 + erow[e], N - 1);
  auto fetch = [&](int step_, Ops& o) {
    const int fs = T - 1 - step_;
    const int t_ = d.reverse ? (T - 1 - fs) : fs;
    const int tp_ = min(max(d.reverse ? t_ + 1 : t_ - 1, 0), T - 1);
#pragma unroll
    for (int e = 0; e < NE; ++e) {
      const int64_t n = nrow[e];
      const int j = ej[e];
#pragma unroll
      for (int g = 0; g < 4; ++g) o.gt[e][g] = d.gates[((int64_t)t_ * N + n) * a.ldg + g * H + j];
      o.cc[e] = d.c_all[((int64_t)t_ * N + n) * H + j];
      const float cpv = d.c_all[((int64_t)tp_ * N + n) * H + j];
      o.cp[e] = fs > 0 ? cpv : 0.f;
      o.dho[e] = d.dh_out[((int64_t)t_ * N + n) * a.ldh + j];
    }
  };
  auto frame = [&](int step, Ops& cur, Ops& nxt) {
    const int fstep = T - 1 - step;
    const int t = d.reverse ? (T - 1 - fstep) : fstep;
    auto& gt = cur.gt;
    auto& cc = cur.cc;
    auto& cp = cur.cp;
    auto& dho = cur.dho;
    fetch(min(step + 1, T - 1), nxt);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if constexpr (B16) {
#pragma unroll
      for (int kc = 0; kc < KW / 32; ++kc) {
        bf16x8 av[NP];
#pragma unroll
        for (int q = 0; q < NP; ++q) av[q] = *reinterpret_cast<const bf16x8*>(&dgs16[q][r][kp * KW + kc * 32 + 8 * kq]);
        constexpr int ia[6] = {2, 1, 0, 1, 0, 0}, ib[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
        for (int term = (NP == 3 ? 0 : 5); term < 6; ++term)
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[ia[term]], wb[ib[term]][kc], acc, 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int kc = 0; kc < 8; ++kc) {
        const f32x4 av = *reinterpret_cast<const f32x4*>(&dgs[r][kp * KW + kc * 16 + 4 * kq]);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[e], wf[kc][e], acc, 0, 0, 0);
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) rs[kp][kq * 4 + q][nt * 16 + r] = acc[q];
    __syncthreads();
#pragma unroll
    for (int e = 0; e < NE; ++e) {
      const int row = erow[e], j = ej[e];
      float dh = dho[e];
#pragma unroll
      for (int p = 0; p < KP; ++p) dh += rs[p][row][j];
      const float gi = gt[e][0], gf = gt[e][1], gg = gt[e][2], go = gt[e][3];
      const float tc = gate_tanh(cc[e]);
      const float dc = dcreg[e] + dh * go * (1.f - tc * tc);
      const float o0 = dc * gg * gi * (1.f - gi);
      const float o1 = dc * cp[e] * gf * (1.f - gf);
      const float o2 = dc * gi * (1.f - gg * gg);
      const float o3 = dh * tc * go * (1.f - go);
      dcreg[e] = dc * gf;
      if constexpr (B16) {
        const float ov[4] = {o0, o1, o2, o3};
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float v = eok[e] ? ov[g] : 0.f;
          if constexpr (PM == 2) {
            __bf16 pl[3];
            split3s(v, pl);
#pragma unroll
            for (int q = 0; q < 3; ++q) dgs16[q][row][g * H + j] = pl[q];
          } else {
            dgs16[0][row][g * H + j] = (__bf16)v;
          }
        }
      } else {
        dgs[row][j] = eok[e] ? o0 : 0.f;
        dgs[row][H + j] = eok[e] ? o1 : 0.f;
        dgs[row][2 * H + j] = eok[e] ? o2 : 0.f;
        dgs[row][3 * H + j] = eok[e] ? o3 : 0.f;
      }
      if (eok[e]) {
        float* o = d.dgates + ((int64_t)t * N + m0 + row) * a.ldg + j;
        o[0] = o0;
        o[H] = o1;
        o[2 * H] = o2;
        o[3 * H] = o3;
      }
    }
    __syncthreads();
  };
  Ops oa, ob;
  fetch(0, oa);
  for (int step = 0; step < T; step += 2) {
    frame(step, oa, ob);
    if (step + 1 < T) frame(step + 1, ob, oa);
  }
}

// (the fragment packs of W_hh are produced by repack.hip: dvae_lstm_pack_w / dvae_repack_all)

int fill_args(StepArgs& a, const dvae_lstm_dir_t* dirs, int ndir, int T, int N, int H, int64_t ldh, bool bwd) {
  if (!dirs || ndir < 1 || ndir > 2 || T < 1 || N < 1 || H < 64 || (H & 63) || (ldh & 3)) return DVAE_EINVAL;
  for (int i = 0; i < ndir; ++i) {
    const dvae_lstm_dir_t& s = dirs[i];
    if (!s.gates || !s.w_hh || !s.c_all) return DVAE_EINVAL;
    if (bwd ? (!s.dh_out || !s.dgates || !s.dc_ws) : (!s.h_out)) return DVAE_EINVAL;
    if ((((uintptr_t)s.w_hh) | ((uintptr_t)s.h_out) | ((uintptr_t)s.dgates)) & 15) return DVAE_EINVAL;
    a.d[i].gates = s.gates; a.d[i].w = s.w_hh; a.d[i].wp = s.w_packed; a.d[i].h_out = s.h_out; a.d[i].c_all = s.c_all;
    a.d[i].dh_out = s.dh_out; a.d[i].dgates = s.dgates; a.d[i].dc = s.dc_ws; a.d[i].reverse = s.reverse;
    a.d[i].shift = s.step_shift;
    if (s.step_shift < 0) return DVAE_EINVAL;
  }
  if (ndir == 1) a.d[1] = a.d[0];
  a.pm = dirs[0].packed_mode;
  if (a.pm != DVAE_MODE_F32 && a.pm != DVAE_MODE_BF16 && a.pm != DVAE_MODE_F32X3) return DVAE_EINVAL;
  for (int i = 0; i < ndir; ++i)
    if (dirs[i].packed_mode != a.pm || (a.pm && H != 64 && !dirs[i].w_packed)) return DVAE_EINVAL;
  // bf16 / fp32x3 frame kernels exist for H = 512, 1024, ...; the H = 64 kernels round / split W_hh themselves: no pack
  if (a.pm && (H % 512) && H != 64) return DVAE_EINVAL;
  a.st16 = dirs[0].state_bf16 ? 1 : 0;
  for (int i = 0; i < ndir; ++i)
    if ((dirs[i].state_bf16 ? 1 : 0) != a.st16) return DVAE_EINVAL;
  if (a.st16 && a.pm != DVAE_MODE_BF16) return DVAE_EINVAL;
  a.T = T; a.N = N; a.H = H; a.ldh = ldh;
  a.ldg = dirs[0].gate_ld ? dirs[0].gate_ld : 4 * H;
  for (int i = 0; i < ndir; ++i)
    if ((dirs[i].gate_ld ? dirs[i].gate_ld : 4 * H) != a.ldg) return DVAE_EINVAL;
  // (only the whole-sequence kernels of H = 64 take a stride; 16-byte rows)
  if (a.ldg != 4 * H && (H != 64 || a.ldg < 4 * H || (a.ldg & 3))) return DVAE_EINVAL;
  return DVAE_OK;
}

}  // namespace

// lstm_pers.hip: the W_hh-resident persistent recurrence (one launch per sequence)
int dvae_pers_usable(int N, int H, int pm, int bwd);
int dvae_pers_bwd_ksplit(int H);
int dvae_pers_fwd_units(int N, int H);
int dvae_pers_launch(const dvae_lstm_dir_t& d, bool bwd, int T, int N, int H, int64_t ldh, int drop_bid, hipStream_t s);

namespace {
// Kernel families, chosen by H (and, for the persistent one, by the caller handing in a workspace) (every one of them is reached by tests/test_hip_kernels.py::test_lstm_layer):
//   H == 64            lstm_seq_*_h64      whole sequence in one launch, W_hh in registers
//   H % 512 == 0       lstm_step_*_v5      one launch per frame, fragment-packed W_hh (needs dirs[i].w_packed)
//   other H % 64 == 0  lstm_step_*_kernel  one launch per frame, W_hh staged through LDS (generic fallback)
struct SeqPlan {
  bool shifted, whole;
  int n_j, mt5, n_m5;
  int64_t frames;   // (entry, step) pairs of this call that carry a recurrent product
};

// dvae_prof_collect_tags, family 2: tag = kind | H-class << 4 | precision mode << 8
//   kind 0 / 1: one launch per frame, forward / backward; 2 / 3: the W_hh-resident persistent launch; 4 / 5: H = 64, whole
//   sequence per launch.  H-class: 0 H = 64, 1 H = 512, 2 H = 1024, 3 other.
// bytes = ALGORITHMIC L2 -> CU operand bytes of the call at its tiling: per frame, every workgroup pulls its slice of W_hh
// (per-frame kernels only) and its rows of h[t-1] / dG[t+1]; the persistent launches pull the rows only.
unsigned lstm_tag(int kind, int H, int pm) {
  const int hc = H == 64 ? 0 : H == 512 ? 1 : H == 1024 ? 2 : 3;
  return (unsigned)kind | ((unsigned)hc << 4) | ((unsigned)pm << 8);
}
double lstm_bytes(bool bwd, bool pers, const StepArgs& a, const SeqPlan& p, int ndir) {
  const double H = a.H, K = bwd ? 4.0 * H : H;             // contraction depth
  const double bw = a.pm == DVAE_MODE_BF16 ? 2.0 : a.pm == DVAE_MODE_F32X3 ? 6.0 : 4.0;
  double ba = (a.pm == DVAE_MODE_BF16 && a.st16) ? 2.0 : 4.0;   // bytes per element of the streamed rows
  if (H == 64) return 0.0;
  if (pers) {
    if (a.pm == DVAE_MODE_F32X3) {
      const double wgs = (H / 16) * ((a.N + 31) / 32);
      // forward: 16 units x 32 rows per workgroup, h[t-1] as three bf16 planes; backward: dG[t+1] in fp32 — all of K = 4H per
      // workgroup, or (k-split kernel) a quarter of it plus the three 32 x 16 partial tiles a workgroup receives
      if (!bwd) return (double)p.frames * wgs * (16.0 / dvae_pers_fwd_units(a.N, a.H)) * 32.0 * K * 6.0;   // (8-unit kernel: twice the workgroups)
      if (dvae_pers_bwd_ksplit(a.H)) return (double)p.frames * wgs * (32.0 * (K / 4.0) * 4.0 + 3.0 * 2048.0);
      return (double)p.frames * wgs * 32.0 * K * 4.0;
    }
    const int mt = ((int)(H / 32) * ((a.N + 15) / 16) <= 256) ? 1 : 2;
    const double wgs = (H / 32) * ((a.N + 16 * mt - 1) / (16 * mt));
    return (double)p.frames * wgs * 16.0 * mt * K * 2.0;     // rows handed over as bf16
  }
  // per-frame kernels: a workgroup = 16 hidden units x 16*mt5 rows; its W_hh slice is 64 x H values in both passes
  const double wgs = (double)p.n_j * p.n_m5;
  (void)ndir;
  if (a.pm == DVAE_MODE_F32 || bwd) ba = (a.pm == DVAE_MODE_BF16 && a.st16) ? 2.0 : 4.0;
  return (double)p.frames * wgs * (64.0 * H * bw + 16.0 * p.mt5 * K * ba);
}
int plan_seq(const StepArgs& a, int ndir, int g0, int g1, SeqPlan& p) {
  if (g0 < 0 || g1 < g0) return DVAE_EINVAL;
  p.shifted = a.d[0].shift != 0 || (ndir == 2 && a.d[1].shift != 0);
  p.whole = (g0 == 0 && g1 == a.T && !p.shifted);
  p.n_j = a.H / 16;
  // 32-row tiles when that still gives every CU a workgroup, else 16-row tiles: either way two waves per SIMD
  p.mt5 = (p.n_j * ((a.N + 31) / 32) * ndir >= 256) ? 2 : 1;
  p.n_m5 = (a.N + 16 * p.mt5 - 1) / (16 * p.mt5);
  p.frames = 0;
  for (int i = 0; i < ndir; ++i)
    for (int g = g0; g < g1; ++g) p.frames += (g - a.d[i].shift > 0 && g - a.d[i].shift < a.T) ? 1 : 0;
  return DVAE_OK;
}

// global launch steps [g0, g1): entry i runs its local step g - step_shift when that lies in [0, T)
int lstm_seq_fwd_range(const dvae_lstm_dir_t* dirs, int ndir, int T, int N, int H, int64_t ldh, int g0, int g1,
                       void* stream) {
  StepArgs a{};
  int rc = fill_args(a, dirs, ndir, T, N, H, ldh, false);
  if (rc) return rc;
  SeqPlan p;
  if ((rc = plan_seq(a, ndir, g0, g1, p))) return rc;
  hipStream_t s = (hipStream_t)stream;
  const bool pers = ndir == 1 && p.whole && dirs[0].pers_ws && dvae_pers_usable(N, H, a.pm, 0);
  ProfScope prof(2, s, 2.0 * N * 4.0 * H * H * (double)p.frames, lstm_tag(H == 64 ? 4 : pers ? 2 : 0, H, a.pm),
                 lstm_bytes(false, pers, a, p, ndir));
  if (pers) return dvae_pers_launch(dirs[0], false, T, N, H, ldh, -1, s);
  if (H == 64) {
    if (!p.whole) return DVAE_EINVAL;
    // rows per workgroup: as few as still leave <= 64 workgroups (N = 128 segments x 2 directions: 4 rows)
    static const int rows_env = dvae_dev_knob("DVAE_LSTM_H64_ROWS", 0);
    int rows = 16;
    if (a.pm) {
      if (((N + 7) / 8) * ndir <= 64) rows = 8;
      if (((N + 3) / 4) * ndir <= 64) rows = 4;
      if (rows_env == 16 || rows_env == 8 || rows_env == 4) rows = rows_env;
    }
#define H64F(PM_, R_) hipLaunchKernelGGL((lstm_seq_fwd_h64<PM_, R_>), dim3((N + R_ - 1) / R_, ndir), dim3(64 * R_), 0, s, a)
    if (a.pm == DVAE_MODE_BF16) { if (rows == 4) H64F(1, 4); else if (rows == 8) H64F(1, 8); else H64F(1, 16); }
    else if (a.pm == DVAE_MODE_F32X3) { if (rows == 4) H64F(2, 4); else if (rows == 8) H64F(2, 8); else H64F(2, 16); }
    else hipLaunchKernelGGL(lstm_seq_fwd_h64<0>, dim3((N + 15) / 16, ndir), dim3(512), 0, s, a);
#undef H64F
    return dvae_check_launch();
  }
  if (H % 512 == 0) {
    if (!a.d[0].wp || !a.d[ndir - 1].wp) return DVAE_EINVAL;
    dim3 grid5(p.n_j * p.n_m5, 1, ndir), block5(512);
    // stacked entries (a shift): the 64-deep variant keeps TWO workgroups resident per CU, so the two layers' frames
    // really overlap (one's load latency under the other's MFMAs) instead of alternating
    for (int step = g0; step < g1; ++step) {
#define FWD5(MT_, KR_, PM_) hipLaunchKernelGGL((lstm_step_fwd_v5<MT_, KR_, PM_, 2>), grid5, block5, 0, s, a, step, p.n_j, p.n_m5)
      if (a.pm == DVAE_MODE_BF16 && a.st16) {
        if (p.mt5 == 2) hipLaunchKernelGGL((lstm_step_fwd_v5<2, 128, 1, 2, true>), grid5, block5, 0, s, a, step, p.n_j, p.n_m5);
        else hipLaunchKernelGGL((lstm_step_fwd_v5<1, 128, 1, 2, true>), grid5, block5, 0, s, a, step, p.n_j, p.n_m5);
      } else if (a.pm == DVAE_MODE_BF16) {
        if (p.mt5 == 2) FWD5(2, 128, 1); else FWD5(1, 128, 1);
      } else if (a.pm == DVAE_MODE_F32X3) {
        // (64-row tiles, which read W_hh twice per frame instead of 4x, measured SLOWER: 15.3 vs 12.5 us per layer-frame
        // for the two stacked H = 1024 layers — one workgroup per CU hides less latency than the bytes it saves)
        if (p.mt5 == 2) FWD5(2, 64, 2); else FWD5(1, 64, 2);
      } else if (p.mt5 == 2 && p.shifted) FWD5(2, 64, 0);
      else if (p.mt5 == 2) FWD5(2, 128, 0);
      else FWD5(1, 64, 0);
#undef FWD5
    }
    return dvae_check_launch();
  }
  if (!p.whole || a.pm) return DVAE_EINVAL;   // step ranges / stacked entries / bf16 / fp32x3 exist for the eight-wave kernels only
  const int n_m = (N + 15) / 16;
  for (int step = 0; step < T; ++step)
    hipLaunchKernelGGL((lstm_step_fwd_kernel<1>), dim3(p.n_j * n_m, 1, ndir), dim3(256), 0, s, a, step, p.n_j, n_m);
  return dvae_check_launch();
}
}  // namespace

DVAE_API int dvae_lstm_seq_fwd(const dvae_lstm_dir_t* dirs, int ndir, int T, int N, int H, int64_t ldh, void* stream) {
  return lstm_seq_fwd_range(dirs, ndir, T, N, H, ldh, 0, T, stream);
}
DVAE_API int dvae_lstm_seq_fwd_range(const dvae_lstm_dir_t* dirs, int ndir, int T, int N, int H, int64_t ldh,
                                     int step_begin, int step_end, void* stream) {
  return lstm_seq_fwd_range(dirs, ndir, T, N, H, ldh, step_begin, step_end, stream);
}

namespace {
int lstm_seq_bwd_range(const dvae_lstm_dir_t* dirs, int ndir, int T, int N, int H, int64_t ldh, int g0, int g1,
                       void* stream) {
  StepArgs a{};
  int rc = fill_args(a, dirs, ndir, T, N, H, ldh, true);
  if (rc) return rc;
  SeqPlan p;
  if ((rc = plan_seq(a, ndir, g0, g1, p))) return rc;
  hipStream_t s = (hipStream_t)stream;
  const bool pers = ndir == 1 && p.whole && dirs[0].pers_ws && dvae_pers_usable(N, H, a.pm, 1);
  ProfScope prof(2, s, 2.0 * N * 4.0 * H * H * (double)p.frames, lstm_tag(H == 64 ? 5 : pers ? 3 : 1, H, a.pm),
                 lstm_bytes(true, pers, a, p, ndir));
  if (pers) return dvae_pers_launch(dirs[0], true, T, N, H, ldh, -1, s);
  if (H == 64) {
    if (!p.whole) return DVAE_EINVAL;
    static const int rows_env = dvae_dev_knob("DVAE_LSTM_H64_ROWS", 0);
    int rows = 16;
    if (a.pm) {
      if (((N + 7) / 8) * ndir <= 64) rows = 8;
      if (((N + 3) / 4) * ndir <= 64) rows = 4;
      if (rows_env == 16 || rows_env == 8 || rows_env == 4) rows = rows_env;
    }
#define H64B(PM_, R_) hipLaunchKernelGGL((lstm_seq_bwd_h64<PM_, R_>), dim3((N + R_ - 1) / R_, ndir), dim3(64 * R_), 0, s, a)
    if (a.pm == DVAE_MODE_BF16) { if (rows == 4) H64B(1, 4); else if (rows == 8) H64B(1, 8); else H64B(1, 16); }
    else if (a.pm == DVAE_MODE_F32X3) { if (rows == 4) H64B(2, 4); else if (rows == 8) H64B(2, 8); else H64B(2, 16); }
    else hipLaunchKernelGGL(lstm_seq_bwd_h64<0>, dim3((N + 15) / 16, ndir), dim3(512), 0, s, a);
#undef H64B
    return dvae_check_launch();
  }
  if (H % 512 == 0) {
    if (!a.d[0].wp || !a.d[ndir - 1].wp) return DVAE_EINVAL;
    dim3 grid5(p.n_j * p.n_m5, 1, ndir), block5(512);
    for (int step = g0; step < g1; ++step) {
#define BWD5(MT_, KR_, PM_) hipLaunchKernelGGL((lstm_step_bwd_v5<MT_, KR_, PM_>), grid5, block5, 0, s, a, step, p.n_j, p.n_m5)
      if (a.pm == DVAE_MODE_BF16 && a.st16) {
        if (p.mt5 == 2) hipLaunchKernelGGL((lstm_step_bwd_v5<2, 64, 1, 1, true>), grid5, block5, 0, s, a, step, p.n_j, p.n_m5);
        else hipLaunchKernelGGL((lstm_step_bwd_v5<1, 64, 1, 1, true>), grid5, block5, 0, s, a, step, p.n_j, p.n_m5);
      } else if (a.pm == DVAE_MODE_BF16) {
        if (p.mt5 == 2) BWD5(2, 64, 1); else BWD5(1, 64, 1);
      } else if (a.pm == DVAE_MODE_F32X3) {
        if (p.mt5 == 2) BWD5(2, 32, 2); else BWD5(1, 64, 2);
      } else if (p.mt5 == 2) {
        if (dvae_dev_knob("DVAE_LSTM_BWD_KR", 64) == 32) BWD5(2, 32, 0); else BWD5(2, 64, 0);
      } else if (dvae_dev_knob("DVAE_LSTM_BWD_KR", 64) == 32) BWD5(1, 32, 0);
      else BWD5(1, 64, 0);
#undef BWD5
    }
    return dvae_check_launch();
  }
  if (!p.whole || a.pm) return DVAE_EINVAL;
  const int n_m = (N + 15) / 16;
  for (int step = 0; step < T; ++step)
    hipLaunchKernelGGL((lstm_step_bwd_kernel<1>), dim3(p.n_j * n_m, 1, ndir), dim3(256), 0, s, a, step, p.n_j, n_m);
  return dvae_check_launch();
}
}  // namespace

DVAE_API int dvae_lstm_seq_bwd(const dvae_lstm_dir_t* dirs, int ndir, int T, int N, int H, int64_t ldh, void* stream) {
  return lstm_seq_bwd_range(dirs, ndir, T, N, H, ldh, 0, T, stream);
}
DVAE_API int dvae_lstm_seq_bwd_range(const dvae_lstm_dir_t* dirs, int ndir, int T, int N, int H, int64_t ldh,
                                     int step_begin, int step_end, void* stream) {
  return lstm_seq_bwd_range(dirs, ndir, T, N, H, ldh, step_begin, step_end, stream);
}
