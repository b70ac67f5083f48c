// W_hh-RESIDENT persistent LSTM recurrence for gfx950 (H = 512 / 1024, bf16 compute mode): ONE launch walks all T frames
// of a layer.  Replaces the one-launch-per-frame kernels of lstm.hip where it applies (nn.LSTM at
// /root/reference/model/disentangled_vae.py:172,193 used at :238,246).
//
// Why: a frame launch re-reads its W_hh slice from L2 every frame (128 / 192 MB of L2 -> CU traffic per H = 1024
// layer-frame) and pays ~4.5 us of launch boundary + cold first tiles; 10-19 ms of a 24-38 ms training step.
//
// Work split: a workgroup (4 waves, one per SIMD, ONE workgroup per CU) owns 32 hidden units x 16*MT mel segments for the
// whole sequence.  Its W_hh slice — forward: the 128 gate columns of its units over K = H; backward: its 32 columns of
// W_hh over K = 4H — is 64 K bf16 values per wave = 256 VGPRs per lane, loaded ONCE from the fragment packs of
// repack.hip.  Wave w contracts over the k-quarter [w*H/4, (w+1)*H/4) of the hidden units (all four gates of them in
// the backward pass); the four partial tiles meet in LDS, the fused gate / cell (or gate-derivative) update follows,
// and the cell state (backward: the dc carry) never leaves registers.
//
// The frame-to-frame dependency crosses workgroups: frame t needs ALL of h[t-1] (backward: dG[t+1]) of its 16*MT rows.
// The (H/32) workgroups that share a row block form a GROUP; each publishes its 32 units of h[t] (bf16, already in
// MFMA A-fragment order: 1 KiB per 16 rows x 32 k) into a two-slot exchange ring and then raises its flag:
//   producer : payload with write-through (sc1) 16-byte stores by ONE wave -> that wave's s_waitcnt vmcnt(0) -> ONE lane
//              stores flag = frames published (sc1);
//   consumer : wave w polls (relaxed, sc1 loads, one lane per producer) ONLY the KW producers of its own k-quarter,
//              then loads their fragments with sc1 loads straight into MFMA operand registers (no LDS staging).
// This is the "flag" hand-off of MI355X_MICROARCH.md (visibility table, first row): every payload byte stored sc1 and
// drained before the flag, every load of it sc1, so no acquire fence is needed; nothing depends on placement.
// Slot reuse is safe with two slots: a workgroup can publish frame t only after it has consumed ALL of frame t-1, which
// every group member published only after it had finished reading frame t-2's slot.
//
// Never an unbounded wait: every poll gives up after `timeout` ticks of the 100 MHz s_memrealtime clock, writes a sticky
// error record (dvae_lstm_pers_check turns it into DVAE_ELAUNCH) and the whole workgroup leaves the frame loop; its
// neighbours then time out on it in turn.  All workgroups are co-resident by construction: grid <= CU count (checked on
// the host), one workgroup per CU (>= 84 KB of LDS each), so a hand-off can only stall behind foreign work on the GPU.
#include <type_traits>
#include "common.h"

namespace {

typedef unsigned u32x4v __attribute__((__vector_size__(16)));

constexpr int PERS_FLAG_LD = 64;              // flag words per row group (H/32 <= 32 producers, padded to 256 B)
constexpr int PERS_MAX_RB = 16;               // row groups
constexpr int PERS_FLAG_BYTES = PERS_MAX_RB * PERS_FLAG_LD * 4;   // 4 KiB, zeroed by a memset node before every launch
constexpr int PERS_ERR_OFF = PERS_FLAG_BYTES; // sticky error record: 16 words (never cleared by a launch)
constexpr int PERS_XCH_OFF = 8192;            // exchange ring
constexpr int PERS_PAD_LDS = 84 * 1024;       // total LDS per workgroup >= this: exactly one workgroup fits a CU

struct PersArgs {
  float* gates;          // [T,N,4H]
  const char* wp;        // fragment pack (forward: packed_fwd, backward: packed_bwd), bf16
  char* h_out;           // forward: [T,N,ldh] bf16 (S16) or fp32
  float* c_all;          // [T,N,H]
  const float* dh_out;   // backward: [T,N,ldh]
  char* dgates;          // backward: [T,N,4H] bf16 (S16) or fp32
  unsigned* flags;       // ws + 0
  unsigned* err;         // ws + PERS_ERR_OFF
  char* xch;             // ws + PERS_XCH_OFF
  int T, N;
  int64_t ldh;
  int reverse, n_rb;
  unsigned timeout;      // 100 MHz ticks
  int xch_bytes;
  int drop_bid;          // self-test: this workgroup never publishes (-1: none)
};

// wave-level bounded poll: lanes with `active` re-read their word until every one of them has reached `target`
__device__ __forceinline__ bool poll_ge(const unsigned* p, bool active, unsigned target, unsigned timeout) {
  unsigned v = active ? __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : target;
  if (__all(v >= target)) return true;
  const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
  for (unsigned it = 1;; ++it) {
    __builtin_amdgcn_s_sleep(1);
    v = active ? __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : target;
    if (__all(v >= target)) return true;
    if ((it & 15) == 0 && (__builtin_amdgcn_s_memrealtime() - t0) > timeout) return false;
  }
}

__device__ __forceinline__ void pers_give_up(unsigned* err, int code, int bid, int step, int wave) {
  if ((threadIdx.x & 63) == 0) {
    if (atomicCAS(err, 0u, (unsigned)code) == 0u) {
      __hip_atomic_store(err + 1, (unsigned)bid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(err + 2, (unsigned)step, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(err + 3, (unsigned)wave, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// ======================================================================================================================
// forward:  G = Xproj[t] + h[t-1] W_hh^T ; i,f,o = sigmoid, g = tanh ; c = f c' + i g ; h = o tanh(c)
// ======================================================================================================================
template <int H, int MT, bool S16>
__global__ __launch_bounds__(256, 1) void lstm_pers_fwd_bf16(const PersArgs a) {
  constexpr int NCH = H / 32;           // 32-deep k-chunks of h = producers of a row group
  constexpr int KW = NCH / 4;           // chunks (= producers) per wave
  constexpr int NEL = 2 * MT;           // (segment, unit) elements per thread
  constexpr int PD = (KW * MT > 8) ? KW / 2 : KW;   // chunks of h[t-1] in flight (registers: 256 hold W_hh)
  using h_t = typename std::conditional<S16, __bf16, float>::type;
  const int T = a.T, N = a.N;
  const int bid = blockIdx.x;
  const int rb = bid % a.n_rb, jb = bid / a.n_rb;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, q = lane >> 4;

  extern __shared__ __attribute__((aligned(16))) char lds_raw[];
  float (*red)[MT * 32][64] = reinterpret_cast<float (*)[MT * 32][64]>(lds_raw);            // [wave][(mt,g,u,e)][lane]
  __bf16 (*hx)[16][40] = reinterpret_cast<__bf16 (*)[16][40]>(lds_raw + 4 * MT * 32 * 64 * 4);   // [mt][row][32 units + pad]
  volatile int* dead = reinterpret_cast<volatile int*>(lds_raw + 4 * MT * 32 * 64 * 4 + MT * 16 * 40 * 2);
  if (tid == 0) *dead = 0;

  // resident W_hh fragments: gate g, 16-unit tile u, chunk k of this wave's k-quarter
  bf16x8 W[4][2][KW];
#pragma unroll
  for (int g = 0; g < 4; ++g)
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int k = 0; k < KW; ++k)
        W[g][u][k] = *reinterpret_cast<const bf16x8*>(
            a.wp + ((((int64_t)(g * (H / 16) + 2 * jb + u)) * NCH + wave * KW + k) * 64 + lane) * 16);

  // this thread's elements: accumulator-layout positions (mt, u, e) of the 16*MT x 32 tile
  int el_row[NEL], el_unit[NEL], el_red[NEL];
  int64_t el_n[NEL];
  bool el_ok[NEL];
  float creg[NEL];
#pragma unroll
  for (int i = 0; i < NEL; ++i) {
    const int s = wave * NEL + i, tile = s >> 2, e = s & 3, mt = tile >> 1, u = tile & 1;
    el_row[i] = mt * 16 + q * 4 + e;
    el_unit[i] = u * 16 + r;
    el_red[i] = (mt * 8 + u) * 4 + e;          // + g*8 (gate stride) inside red
    const int n = rb * 16 * MT + el_row[i];
    el_ok[i] = n < N;
    el_n[i] = min(n, N - 1);
    creg[i] = 0.f;
  }
  const int j0 = jb * 32;
  const int64_t H4 = 4 * (int64_t)H;

  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)a.xch, 0, a.xch_bytes, 0x00020000);
  const int slot_bytes = a.n_rb * NCH * MT * 1024;
  const int xld = (rb * NCH + wave * KW) * MT * 1024 + lane * 16;      // this wave's first fragment inside a slot
  const int xst = (rb * NCH + jb) * MT * 1024 + lane * 16;             // where this workgroup publishes
  const unsigned* pflag = a.flags + rb * PERS_FLAG_LD + wave * KW + (lane < KW ? lane : 0);
  unsigned* myflag = a.flags + rb * PERS_FLAG_LD + jb;

  auto fetch = [&](int step_, float (&x)[NEL][4]) {
    const int t_ = a.reverse ? (T - 1 - step_) : step_;
    const float* __restrict__ G_ = a.gates + (int64_t)t_ * N * H4 + j0;
#pragma unroll
    for (int i = 0; i < NEL; ++i)
#pragma unroll
      for (int g = 0; g < 4; ++g) x[i][g] = G_[el_n[i] * H4 + g * H + el_unit[i]];
  };

  auto frame = [&](int step, float (&xp)[NEL][4], float (&xn)[NEL][4]) -> bool {
    const int t = a.reverse ? (T - 1 - step) : step;
    f32x4 acc[MT][4][2];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int u = 0; u < 2; ++u) acc[mt][g][u] = f32x4{0.f, 0.f, 0.f, 0.f};

    if (step > 0) {
      if (!poll_ge(pflag, lane < KW, (unsigned)step, a.timeout)) {
        pers_give_up(a.err, 1, bid, step, wave);
        *dead = 1;
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");      // no instruction: keeps the loads below the poll
      const int so = ((step - 1) & 1) * slot_bytes;
      bf16x8 av[PD][MT];                      // PD chunks of h[t-1] in flight
      auto load = [&](int k) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
          av[k % PD][mt] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(xrs, xld + (k * MT + mt) * 1024, so, 16));
      };
#pragma unroll
      for (int k = 0; k < PD; ++k) load(k);
      fetch(min(step + 1, T - 1), xn);      // next frame's pre-activations, in flight under this frame
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int k = 0; k < KW; ++k) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int u = 0; u < 2; ++u)
              acc[mt][g][u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[k % PD][mt], W[g][u][k], acc[mt][g][u], 0, 0, 0);
        if (k + PD < KW) {
          __builtin_amdgcn_sched_barrier(0);
          load(k + PD);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    } else {
      fetch(min(step + 1, T - 1), xn);
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int e = 0; e < 4; ++e) red[wave][((mt * 4 + g) * 2 + u) * 4 + e][lane] = acc[mt][g][u][e];
    __syncthreads();
    if (*dead) return false;

    float* __restrict__ G = a.gates + (int64_t)t * N * H4 + j0;
#pragma unroll
    for (int i = 0; i < NEL; ++i) {
      float gsum[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int ri = el_red[i] + g * 8;
        gsum[g] = (red[0][ri][lane] + red[1][ri][lane]) + (red[2][ri][lane] + red[3][ri][lane]) + xp[i][g];
      }
      const float gi = gate_sigmoid(gsum[0]);
      const float gf = gate_sigmoid(gsum[1]);
      const float gg = gate_tanh(gsum[2]);
      const float go = gate_sigmoid(gsum[3]);
      const float c = gf * creg[i] + gi * gg;
      const float h = go * gate_tanh(c);
      creg[i] = c;
      hx[el_row[i] >> 4][el_row[i] & 15][el_unit[i]] = (__bf16)h;
      if (el_ok[i]) {
        float* g = G + el_n[i] * H4 + el_unit[i];
        g[0] = gi;
        g[H] = gf;
        g[2 * H] = gg;
        g[3 * H] = go;
        a.c_all[((int64_t)t * N + el_n[i]) * H + j0 + el_unit[i]] = c;
        if constexpr (!S16)
          reinterpret_cast<float*>(a.h_out)[((int64_t)t * N + el_n[i]) * a.ldh + j0 + el_unit[i]] = h;
      }
    }
    __syncthreads();
    if (wave == 0) {
      // lane (r, q) holds units 8q..8q+7 of row r: the 16 bytes of the A fragment AND of the bf16 h_out row
      const bool pub = (step + 1 < T) && (bid != a.drop_bid);
      const int so = (step & 1) * slot_bytes;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(&hx[mt][r][q * 8]);
        if (pub) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4v, v),
                                                        xrs, xst + mt * 1024, so, 16);
        if constexpr (S16) {
          const int n = rb * 16 * MT + mt * 16 + r;
          if (n < N)
            *reinterpret_cast<f32x4*>(reinterpret_cast<h_t*>(a.h_out) + ((int64_t)t * N + n) * a.ldh + j0 + q * 8) = v;
        }
      }
      if (pub) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the write-through payload has left before the flag does
        if (lane == 0) __hip_atomic_store(myflag, (unsigned)(step + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    return true;
  };

  float xa[NEL][4], xb[NEL][4];
  fetch(0, xa);
  __syncthreads();
  for (int step = 0; step < T; step += 2) {
    if (!frame(step, xa, xb)) break;
    if (step + 1 < T && !frame(step + 1, xb, xa)) break;
  }
}

// ======================================================================================================================
// backward:  dH = dHout[t] + dG[t+1] W_hh ; gate derivatives -> dG[t] ; dC carry in registers
// ======================================================================================================================
template <int H, int MT, bool S16>
__global__ __launch_bounds__(256, 1) void lstm_pers_bwd_bf16(const PersArgs a) {
  constexpr int NCH = H / 32;           // chunks per gate = producers of a row group
  constexpr int KW = NCH / 4;
  constexpr int NEL = 2 * MT;
  constexpr int NC = 4 * KW;            // chunks a wave contracts per frame (its unit quarter of all four gates)
  constexpr int PD = (NC >= 16 && MT == 1) ? 8 : 4;   // chunks of dG in flight (registers: 256 hold W_hh)
  using g_t = typename std::conditional<S16, __bf16, float>::type;
  const int T = a.T, N = a.N;
  const int bid = blockIdx.x;
  const int rb = bid % a.n_rb, jb = bid / a.n_rb;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, q = lane >> 4;

  extern __shared__ __attribute__((aligned(16))) char lds_raw[];
  float (*red)[MT * 8][64] = reinterpret_cast<float (*)[MT * 8][64]>(lds_raw);                 // [wave][(mt,u,e)][lane]
  __bf16 (*gx)[16][40] = reinterpret_cast<__bf16 (*)[16][40]>(lds_raw + 4 * MT * 8 * 64 * 4);    // [(g,mt)][row][32 units + pad]
  volatile int* dead = reinterpret_cast<volatile int*>(lds_raw + 4 * MT * 8 * 64 * 4 + 4 * MT * 16 * 40 * 2);
  if (tid == 0) *dead = 0;

  // resident W_hh fragments (rows g*H + k-quarter of this wave, columns = the 32 units of this workgroup)
  bf16x8 W[4][2][KW];
#pragma unroll
  for (int g = 0; g < 4; ++g)
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int k = 0; k < KW; ++k)
        W[g][u][k] = *reinterpret_cast<const bf16x8*>(
            a.wp + ((((int64_t)((2 * jb + u) * 4 + g)) * NCH + wave * KW + k) * 64 + lane) * 16);

  int el_row[NEL], el_unit[NEL], el_red[NEL];
  int64_t el_n[NEL];
  bool el_ok[NEL];
  float dcreg[NEL], ccreg[NEL];
#pragma unroll
  for (int i = 0; i < NEL; ++i) {
    const int s = wave * NEL + i, tile = s >> 2, e = s & 3, mt = tile >> 1, u = tile & 1;
    el_row[i] = mt * 16 + q * 4 + e;
    el_unit[i] = u * 16 + r;
    el_red[i] = (mt * 2 + u) * 4 + e;
    const int n = rb * 16 * MT + el_row[i];
    el_ok[i] = n < N;
    el_n[i] = min(n, N - 1);
    dcreg[i] = 0.f;
  }
  const int j0 = jb * 32;
  const int64_t H4 = 4 * (int64_t)H;

  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)a.xch, 0, a.xch_bytes, 0x00020000);
  const int slot_bytes = a.n_rb * 4 * NCH * MT * 1024;
  const int xld = (rb * 4 * NCH + wave * KW) * MT * 1024 + lane * 16;     // + g*NCH*MT*1024 per gate
  const int xst = (rb * 4 * NCH + jb) * MT * 1024 + lane * 16;
  const unsigned* pflag = a.flags + rb * PERS_FLAG_LD + wave * KW + (lane < KW ? lane : 0);
  unsigned* myflag = a.flags + rb * PERS_FLAG_LD + jb;

  struct Ops {
    float gt[NEL][4], cp[NEL], dho[NEL];
  };
  auto frame_t = [&](int step_) { const int fs = T - 1 - step_; return a.reverse ? (T - 1 - fs) : fs; };
  auto fetch = [&](int step_, Ops& o) {
    const int t_ = frame_t(step_);
    const int tp_ = min(max(a.reverse ? t_ + 1 : t_ - 1, 0), T - 1);
#pragma unroll
    for (int i = 0; i < NEL; ++i) {
#pragma unroll
      for (int g = 0; g < 4; ++g) o.gt[i][g] = a.gates[((int64_t)t_ * N + el_n[i]) * H4 + g * H + j0 + el_unit[i]];
      o.cp[i] = a.c_all[((int64_t)tp_ * N + el_n[i]) * H + j0 + el_unit[i]];
      o.dho[i] = a.dh_out[((int64_t)t_ * N + el_n[i]) * a.ldh + j0 + el_unit[i]];
    }
  };

  auto frame = [&](int step, Ops& cur, Ops& nxt) -> bool {
    const int fstep = T - 1 - step;
    const int t = frame_t(step);
    f32x4 acc[MT][2];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int u = 0; u < 2; ++u) acc[mt][u] = f32x4{0.f, 0.f, 0.f, 0.f};

    if (step > 0) {
      if (!poll_ge(pflag, lane < KW, (unsigned)step, a.timeout)) {
        pers_give_up(a.err, 2, bid, step, wave);
        *dead = 1;
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      const int so = ((step - 1) & 1) * slot_bytes;
      // chunk c of this wave: gate g = c / KW, chunk k = c % KW of its k-quarter; PD chunks in flight
      bf16x8 av[PD][MT];
      auto load = [&](int c) {
        const int g = c / KW, k = c % KW;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
          av[c % PD][mt] = __builtin_bit_cast(
              bf16x8, __builtin_amdgcn_raw_buffer_load_b128(xrs, xld + ((g * NCH + k) * MT + mt) * 1024, so, 16));
      };
#pragma unroll
      for (int c = 0; c < PD; ++c) load(c);
      fetch(min(step + 1, T - 1), nxt);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        const int g = c / KW, k = c % KW;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int u = 0; u < 2; ++u)
            acc[mt][u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[c % PD][mt], W[g][u][k], acc[mt][u], 0, 0, 0);
        if (c + PD < NC) {
          __builtin_amdgcn_sched_barrier(0);
          load(c + PD);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    } else {
      fetch(min(step + 1, T - 1), nxt);
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int e = 0; e < 4; ++e) red[wave][(mt * 2 + u) * 4 + e][lane] = acc[mt][u][e];
    __syncthreads();
    if (*dead) return false;

#pragma unroll
    for (int i = 0; i < NEL; ++i) {
      const int ri = el_red[i];
      const float rec = (red[0][ri][lane] + red[1][ri][lane]) + (red[2][ri][lane] + red[3][ri][lane]);
      const float dh = cur.dho[i] + rec;
      const float gi = cur.gt[i][0], gf = cur.gt[i][1], gg = cur.gt[i][2], go = cur.gt[i][3];
      const float cp = fstep > 0 ? cur.cp[i] : 0.f;
      const float tc = gate_tanh(ccreg[i]);
      const float dc = dcreg[i] + dh * go * (1.f - tc * tc);
      float o[4];
      o[0] = dc * gg * gi * (1.f - gi);
      o[1] = dc * cp * gf * (1.f - gf);
      o[2] = dc * gi * (1.f - gg * gg);
      o[3] = dh * tc * go * (1.f - go);
      dcreg[i] = dc * gf;
      ccreg[i] = cp;                      // c[t-1] is the cell state of the next (earlier) frame
#pragma unroll
      for (int g = 0; g < 4; ++g) gx[g * MT + (el_row[i] >> 4)][el_row[i] & 15][el_unit[i]] = (__bf16)o[g];
      if constexpr (!S16) {
        if (el_ok[i]) {
          float* dst = reinterpret_cast<float*>(a.dgates) + ((int64_t)t * N + el_n[i]) * H4 + j0 + el_unit[i];
#pragma unroll
          for (int g = 0; g < 4; ++g) dst[g * H] = o[g];
        }
      }
    }
    __syncthreads();
    if (wave == 0) {
      const bool pub = (step + 1 < T) && (bid != a.drop_bid);
      const int so = (step & 1) * slot_bytes;
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          const f32x4 v = *reinterpret_cast<const f32x4*>(&gx[g * MT + mt][r][q * 8]);
          if (pub) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4v, v),
                                                          xrs, xst + (g * NCH * MT + mt) * 1024, so, 16);
          if constexpr (S16) {
            const int n = rb * 16 * MT + mt * 16 + r;
            if (n < N)
              *reinterpret_cast<f32x4*>(reinterpret_cast<g_t*>(a.dgates) + ((int64_t)t * N + n) * H4 + g * H + j0 + q * 8) = v;
          }
        }
      if (pub) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_store(myflag, (unsigned)(step + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    return true;
  };

  // the first frame's cell state, then frame operands one frame ahead (two register sets swapping roles)
  {
    const int t0 = frame_t(0);
#pragma unroll
    for (int i = 0; i < NEL; ++i) ccreg[i] = a.c_all[((int64_t)t0 * N + el_n[i]) * H + j0 + el_unit[i]];
  }
  Ops oa, ob;
  fetch(0, oa);
  __syncthreads();
  for (int step = 0; step < T; step += 2) {
    if (!frame(step, oa, ob)) break;
    if (step + 1 < T && !frame(step + 1, ob, oa)) break;
  }
}

int g_pers_cus = -1;
int pers_cu_count() {
  if (g_pers_cus < 0) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = 0;
    g_pers_cus = n;
  }
  return g_pers_cus;
}

// rows per workgroup (16 * MT) such that (H/32) * row blocks fits the chip; 0: shape not supported
int pers_mt(int N, int H, int cus) {
  if (H != 512 && H != 1024) return 0;
  const int n_jb = H / 32;
  for (int mt = 1; mt <= 2; ++mt) {
    const int n_rb = (N + 16 * mt - 1) / (16 * mt);
    if (n_rb <= PERS_MAX_RB && n_jb * n_rb <= cus) return mt;
  }
  return 0;
}

}  // namespace

// used by lstm.hip: 1 when (N, H, mode) has a persistent kernel on this device
int dvae_pers_usable(int N, int H, int pm) {
  if (pm != DVAE_MODE_BF16) return 0;
  return pers_mt(N, H, pers_cu_count()) != 0;
}

DVAE_API int64_t dvae_lstm_pers_ws_bytes(int N, int H) {
  if (N < 1 || (H != 512 && H != 1024)) return 0;
  int mt = pers_mt(N, H, 256);
  if (!mt) return 0;
  const int64_t n_rb = (N + 16 * mt - 1) / (16 * mt);
  return PERS_XCH_OFF + 2 * n_rb * (4 * (H / 32)) * mt * 1024;
}

// one direction of one layer, all T frames; `bwd` selects the pass.  Returns DVAE_EINVAL when the shape has no persistent
// kernel (the caller then uses the per-frame launches).
int dvae_pers_launch(const dvae_lstm_dir_t& d, bool bwd, int T, int N, int H, int64_t ldh, int drop_bid, hipStream_t s) {
  const int cus = pers_cu_count();
  const int mt = pers_mt(N, H, cus);
  if (!mt || !d.pers_ws || d.packed_mode != DVAE_MODE_BF16 || !d.w_packed) return DVAE_EINVAL;
  if (((uintptr_t)d.pers_ws) & 255) return DVAE_EINVAL;
  const bool s16 = d.state_bf16 != 0;
  if (s16 && (ldh & 7)) return DVAE_EINVAL;
  PersArgs a{};
  a.gates = d.gates; a.wp = (const char*)d.w_packed; a.h_out = (char*)d.h_out; a.c_all = d.c_all;
  a.dh_out = d.dh_out; a.dgates = (char*)d.dgates;
  char* ws = (char*)d.pers_ws;
  a.flags = (unsigned*)ws; a.err = (unsigned*)(ws + PERS_ERR_OFF); a.xch = ws + PERS_XCH_OFF;
  a.T = T; a.N = N; a.ldh = ldh; a.reverse = d.reverse;
  a.n_rb = (N + 16 * mt - 1) / (16 * mt);
  const unsigned us = d.pers_timeout_us ? d.pers_timeout_us : 2000000u;
  a.timeout = us > 40000000u ? 4000000000u : us * 100u;
  a.xch_bytes = 2 * a.n_rb * (bwd ? 4 : 1) * (H / 32) * mt * 1024;
  a.drop_bid = drop_bid;
  const int grid = (H / 32) * a.n_rb;
  if (hipMemsetAsync(ws, 0, PERS_FLAG_BYTES, s) != hipSuccess) return dvae_check_launch() ? DVAE_ELAUNCH : DVAE_ELAUNCH;
#define PERS_LAUNCH(K, H_, MT_, S_)                                                                      \
  do {                                                                                                   \
    auto kern = K<H_, MT_, S_>;                                                                          \
    static bool attr_set = false;                                                                        \
    if (!attr_set) {                                                                                     \
      (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, PERS_PAD_LDS); \
      attr_set = true;                                                                                   \
    }                                                                                                    \
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), PERS_PAD_LDS, s, a);                                 \
  } while (0)
#define PERS_DISPATCH(K)                                            \
  do {                                                              \
    if (H == 1024 && mt == 1 && s16) PERS_LAUNCH(K, 1024, 1, true); \
    else if (H == 1024 && mt == 2 && s16) PERS_LAUNCH(K, 1024, 2, true); \
    else if (H == 1024 && mt == 1) PERS_LAUNCH(K, 1024, 1, false);  \
    else if (H == 1024) PERS_LAUNCH(K, 1024, 2, false);             \
    else if (mt == 1 && s16) PERS_LAUNCH(K, 512, 1, true);          \
    else if (mt == 2 && s16) PERS_LAUNCH(K, 512, 2, true);          \
    else if (mt == 1) PERS_LAUNCH(K, 512, 1, false);                \
    else PERS_LAUNCH(K, 512, 2, false);                             \
  } while (0)
  if (bwd) PERS_DISPATCH(lstm_pers_bwd_bf16);
  else PERS_DISPATCH(lstm_pers_fwd_bf16);
#undef PERS_DISPATCH
#undef PERS_LAUNCH
  return dvae_check_launch();
}

DVAE_API int dvae_lstm_pers_check(void* ws, int* info4, void* stream) {
  if (!ws) return DVAE_EINVAL;
  unsigned rec[4] = {0, 0, 0, 0};
  hipStream_t s = (hipStream_t)stream;
  if (hipMemcpyAsync(rec, (char*)ws + PERS_ERR_OFF, sizeof(rec), hipMemcpyDeviceToHost, s) != hipSuccess ||
      hipStreamSynchronize(s) != hipSuccess) {
    g_dvae_last_hip_error = (int)hipGetLastError();
    return DVAE_ELAUNCH;
  }
  if (info4)
    for (int i = 0; i < 4; ++i) info4[i] = (int)rec[i];
  if (rec[0] == 0) return DVAE_OK;
  (void)hipMemsetAsync((char*)ws + PERS_ERR_OFF, 0, 64, s);      // reported: clear the sticky record
  (void)hipStreamSynchronize(s);
  g_dvae_last_hip_error = 0;
  return DVAE_ELAUNCH;
}

// self-test of the bounded spin: a forward launch in which workgroup `drop_bid` never publishes; every workgroup must give
// up within the timeout and dvae_lstm_pers_check must then report DVAE_ELAUNCH
DVAE_API int dvae_lstm_pers_selftest(const dvae_lstm_dir_t* dir, int T, int N, int H, int64_t ldh, int drop_bid, void* stream) {
  if (!dir) return DVAE_EINVAL;
  return dvae_pers_launch(*dir, false, T, N, H, ldh, drop_bid, (hipStream_t)stream);
}
