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
//   consumer : ONE wave polls the group's flags (relaxed, sc1 loads, one lane per producer), a barrier, then every wave
//              loads the fragments of its k-quarter with sc1 loads straight into MFMA operand registers (no LDS staging).
//              (lstm_pers_bwd_bf16 and lstm_pers_fwd_x3h: every wave polls the producers of its OWN k-quarter and goes on
//              without the barrier — measured faster there and slower in the other kernels, see those two.)
// This is the "flag" hand-off of MI355X_MICROARCH.md (visibility table, first row): every payload byte stored sc1 and
// drained before the flag, every load of it sc1, so no acquire fence is needed; nothing depends on placement.
// Slot reuse is safe with two slots: a workgroup can publish frame t only after it has consumed ALL of frame t-1, which
// every group member published only after it had finished reading frame t-2's slot.
//
// Never an unbounded wait: every poll gives up after `timeout` ticks of the 100 MHz s_memrealtime clock, writes a sticky
// error record (dvae_lstm_pers_check turns it into DVAE_ELAUNCH) and the whole workgroup leaves the frame loop; its
// neighbours then time out on it in turn.  All workgroups are co-resident by construction: grid <= CU count (checked on
// the host), one workgroup per CU (>= 84 KB of LDS each), so a hand-off can only stall behind foreign work on the GPU.
#include <algorithm>
#include <type_traits>
#include "common.h"

namespace {

typedef unsigned u32x4v __attribute__((__vector_size__(16)));

// Flags: one 128-byte line per producer (32 producers x 128 B per row group).  Packed into one line (32 producers x 4 B)
// the write-through flag stores of a group and its pollers all hit ONE line and a frame took 6.7 us instead of 3.7
// (H = 1024, N = 128, measured): partial-line stores of 32 CUs serialise behind each other.
constexpr int PERS_FLAG_STRIDE = 32;          // words between the flags of two producers
constexpr int PERS_FLAG_LD = 32 * PERS_FLAG_STRIDE;   // words per row group (H/32 <= 32 producers)
constexpr int PERS_FLAG_LD_X3 = 64 * PERS_FLAG_STRIDE;   // fp32x3 forward: H/16 <= 64 producers per row group
constexpr int PERS_MAX_RB = 16;               // row groups
constexpr int PERS_FLAG_BYTES = PERS_MAX_RB * PERS_FLAG_LD * 4;   // 64 KiB; zeroed ONCE by the caller: no launch clears flags (they carry their epoch, pers_epoch)
static_assert(4 * PERS_FLAG_LD_X3 * 4 <= PERS_FLAG_BYTES, "fp32x3: at most 4 row groups of 64 producers");
constexpr int PERS_ERR_OFF = PERS_FLAG_BYTES; // sticky error record: 16 words (never cleared by a launch).  INVARIANT: words 8
                                              // (the flags' epoch) and 9 (workgroups done) belong to the launches — nobody else may
                                              // write them, dvae_lstm_pers_check clears words 0..7 only: a zeroed epoch beside
                                              // flags that still hold an old epoch's counts would publish frames nobody stored
constexpr int PERS_XCH_OFF = PERS_FLAG_BYTES + 4096;   // exchange ring
constexpr int PERS_PAD_LDS = 84 * 1024;       // total LDS per workgroup >= this: exactly one workgroup fits a CU

struct PersArgs {
  float* gates;          // [T,N,4H]
  const char* wp;        // fragment pack (forward: packed_fwd, backward: packed_bwd), bf16
  char* h_out;           // forward: [T,N,ldh] bf16 (s16) or fp32
  float* c_all;          // [T,N,H]
  const float* dh_out;   // backward: [T,N,ldh]
  char* dgates;          // backward: [T,N,4H] bf16 (s16) or fp32
  float* db1;            // backward, optional: += column sums of dG over all frames and rows (bias gradients)
  float* db2;
  float* dbp;            // backward, optional, instead of db1 / db2: row group rb STORES its share at dbp[(rb * 4 + g) * H + unit]
                         // (one writer per element; the caller adds the n_rb slabs in a fixed order: no atomics)
  unsigned* flags;       // ws + 0
  unsigned* err;         // ws + PERS_ERR_OFF
  char* xch;             // ws + PERS_XCH_OFF
  int T, N;
  int64_t ldh;
  int reverse, n_rb, s16;
  unsigned timeout;      // 100 MHz ticks
  int xch_bytes;
  int drop_bid;          // self-test: this workgroup never publishes (-1: none)
  int local_ok;          // 1: the geometry keeps every row group on ONE XCD when workgroups are dealt round-robin (n_rb % 8 == 0): the
                         // workgroups check that at frame 1 and then keep payload and flags in that XCD's L2 (pers_loc_*)
#ifdef DVAE_PERS_TS
  unsigned long long* ts;   // dev build: [frame][wave][8] s_memrealtime stamps of workgroup ts_bid (scripts/lstm_pers_timeline.py)
  int ts_bid;
#endif
#ifdef DVAE_DEV
  // dev build, fp32x3 forward only (scripts/x3_fwd16_diag2.py): what every consumer had in its registers, frame by frame
  unsigned* dbg;            // [frame][workgroup][thread][12]: xor-fold of the loaded fragments (4), pre-activations used (4), gate sums (4)
  unsigned* dbg_frag;       // [frame][row group][wave][unit*3 + plane][lane][4]: the raw fragments of workgroups with jb == dbg_jb
  int dbg_jb;
  int nslot;                // ring slots (2; T = one per frame: no address is reused inside a launch)
  int nodrain;              // TIMING EXPERIMENT ONLY (results may be wrong): the flag does not wait for the payload's acknowledgement
#endif
};

#ifdef DVAE_PERS_TS
#define PERS_STAMP(p_) do { if (a.ts && bid == a.ts_bid && lane == 0) a.ts[((int64_t)step * 8 + wave) * 8 + (p_)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define PERS_STAMP(p_) do {} while (0)
#endif

#ifdef DVAE_DEV
#define PERS_DRAIN() do { if (!a.nodrain) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); } while (0)
#else
#define PERS_DRAIN() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#endif

// ======================================================================================================================
// XCD-LOCAL hand-off (round 6).  The write-through (sc1) stores of the flag protocol drop their lines from the XCD's L2, so every
// consumer — also one on the producer's XCD — reads flags and fragments at the fabric's latency (MI355X_MICROARCH.md, "stores of
// each flavour").  Where ALL workgroups of a row group sit on ONE XCD, that XCD's L2 is their single point of coherence: plain
// payload stores + `s_waitcnt vmcnt(0)` (acknowledged by the L2) + a workgroup-scope flag store keep the lines there, and the
// consumers' sc1 loads (which bypass the L1 only) hit them: bf16 H = 1024, N = 128: forward 3.24 -> 2.72 us per frame, backward
// 4.03 -> 3.30; H = 512: 2.58 -> 2.19 / 2.97 -> 2.26; fp32x3 H = 512 backward 4.62 -> 3.90 (scripts/local_probe.sh).
// Placement is not promised by HIP, so nothing is assumed:
//   * geometry: workgroups are dealt round-robin over the 8 XCDs (blocks b and b + 8 share one), rb = bid % n_rb: with
//     n_rb % 8 == 0 a row group's members all have the same bid % 8 (host: PersArgs.local_ok);
//   * check: with its FIRST publish (always write-through) a workgroup announces {epoch, XCC_ID} in its flag's line; at frame 1
//     every poller lane compares its producer's announcement with its own XCC_ID.  Every member of a row group evaluates the same
//     predicate over the same announcements, so they agree; any mismatch (or a stale announcement) = write-through as before;
//   * the last two publishes of a launch (one per ring slot) and the last flag store are write-through again, and a workgroup that
//     leaves through a give-up writes its lines through once more: no line of the ring or of a flag stays behind in an L2 for a
//     later launch, whose geometry and placement may differ, to hit.
// ======================================================================================================================
__device__ __forceinline__ unsigned pers_xcc() {
  unsigned x;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
  return x & 0xfu;
}
__device__ __forceinline__ unsigned long long pers_loc_tag(unsigned epoch, unsigned xcc) {
  return ((unsigned long long)epoch << 8) | 0x80u | xcc;      // (0x80: never equal to a word nobody wrote)
}
// the publishing lane, in front of the drain of its first payload: words 2..3 of the flag's own 128-byte line
__device__ __forceinline__ void pers_loc_announce(unsigned* myflag, unsigned long long tag) {
  __hip_atomic_store(reinterpret_cast<unsigned long long*>(myflag + 2), tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// poller lanes (one per awaited producer), after their flags have matched: does every one of them sit on this XCD?
__device__ __forceinline__ bool pers_loc_check(const unsigned* pflag, bool active, unsigned long long tag) {
  const unsigned long long e =
      active ? __hip_atomic_load(reinterpret_cast<const unsigned long long*>(pflag + 2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : tag;
  return __all(e == tag);
}
// publish step `s` of T (frames 0 .. T-2 publish): in the L2 only between the check and the last two publishes
__device__ __forceinline__ bool pers_loc_step(bool loc, int step, int T) { return loc && step >= 1 && step + 4 <= T; }
#define PERS_ST(lp_, v_, rs_, voff_, soff_)                                                \
  do {                                                                                     \
    if (lp_) __builtin_amdgcn_raw_buffer_store_b128(v_, rs_, voff_, soff_, 0);             \
    else __builtin_amdgcn_raw_buffer_store_b128(v_, rs_, voff_, soff_, 16);                \
  } while (0)
#define PERS_FLAG(lp_, p_, val_)                                                                         \
  do {                                                                                                   \
    if (lp_) __hip_atomic_store(p_, val_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);               \
    else __hip_atomic_store(p_, val_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);                       \
  } while (0)
// a workgroup that leaves through a give-up: its flag's line once more with a write-through store (the payload pieces: caller)
__device__ __forceinline__ void pers_loc_scrub_flag(unsigned* myflag) {
  const unsigned v = __hip_atomic_load(myflag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(myflag, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// wave-level bounded poll: lanes with `active` re-read their word until every one of them has reached `target`
// (flags count frames over ALL launches — see pers_epoch — and never wrap)
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

// EPOCH of the flags (round 5: no clearing launch in front of a persistent launch).  A flag holds `epoch + frames published`,
// where `epoch` (word 8 of the error record) is what every workgroup reads when it starts and what the LAST workgroup to
// finish (word 9 counts them) advances by T + 1 — not workgroup 0: a workgroup of another row group may not even have started
// when workgroup 0 is done.  Whatever an earlier launch left in a flag is below the epoch of every later launch.  No wrap:
// once the epoch has passed 2^30 the last workgroup — nobody else is left — zeroes every flag and starts again from 0.
constexpr int PERS_EPOCH_WORD = 8, PERS_DONE_WORD = 9;
constexpr int PERS_LOCAL_WORD = 10;      // statistics: launches whose workgroup 0 went XCD-local (pers_loc_*); read by tests / the selftest
constexpr unsigned PERS_EPOCH_MAX = 1u << 30;
__device__ __forceinline__ unsigned pers_epoch(const PersArgs& a) {
  return __builtin_amdgcn_readfirstlane(__hip_atomic_load(a.err + PERS_EPOCH_WORD, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ void pers_finish(const PersArgs& a) {
  __shared__ int recycle;
  if (threadIdx.x == 0) recycle = 0;
  // Every flag store of this workgroup — also one that was never polled (a give-up path, a dropped workgroup) — must be IN
  // MEMORY before its DONE increment, so that the last workgroup's zeroing of the flags (recycle) cannot be overtaken by a late
  // flag of ~2^30 + T.  Flags are write-through (sc1) stores: once acknowledged they are at the memory side, and this wait
  // (every wave, before the barrier) is that acknowledgement.  (An agent-scope RELEASE on the increment was measured first,
  // its L2 write-back runs over the hundreds of MB of gates / dG this launch has just written — +20 us per launch.)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned d = __hip_atomic_fetch_add(a.err + PERS_DONE_WORD, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (d == gridDim.x - 1) {
      __hip_atomic_store(a.err + PERS_DONE_WORD, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned e = atomicAdd(a.err + PERS_EPOCH_WORD, (unsigned)a.T + 1u) + (unsigned)a.T + 1u;
      if (e > PERS_EPOCH_MAX) recycle = 1;
    }
  }
  __syncthreads();
  if (recycle) {
    __threadfence();      // (once per 1.4 M steps) behind every peer's acknowledged stores, in front of the zeroing
    for (int i = threadIdx.x; i < PERS_FLAG_BYTES / 4; i += blockDim.x)
      __hip_atomic_store(a.flags + i, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(a.err + PERS_EPOCH_WORD, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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

// Frame structure of both kernels (four waves; wave 3 polls, wave 0 publishes, wave 1 archives the bf16 state):
//   [wave 3: bounded poll of the group's flags]  barrier A
//   every wave: sc1 loads of its k-quarter of the handed-over fragments -> MFMAs against its resident W_hh fragments
//               (next frame's epilogue operands are fetched meanwhile) -> partial tiles to LDS (16-byte accesses)  barrier B
//   fused epilogue on 2*MT elements per thread; the new state goes to LDS in fragment order, everything else the frame
//   leaves in HBM goes to an LDS staging tile                                                                     barrier C
//   wave 0: payload (sc1) -> drain -> flag;   meanwhile waves 1..3 (wave 0 joins): whole-line stores of the staging tile
constexpr int NWV = 4;
// fragments in flight at 32-row tiles and H = 1024 (measured)
#ifndef PERS_SPLIT3
#define PERS_SPLIT3 split3      // (a variant with plain, v_pk_add_f32-packed subtractions measured the same: 11.7 vs 11.8 us)
#endif
#ifndef PERS_RD_X3
#define PERS_RD_X3 4
#endif
#ifndef PERS_RD_F32
#define PERS_RD_F32 8
#endif
#ifndef PERS_RD_X3H
#define PERS_RD_X3H 4      // lstm_pers_fwd_x3h: (chunk, row tile) units of h[t-1] in flight, of 8 per wave
#endif
#ifndef PERS_PD_FWD2
#define PERS_PD_FWD2 8      // (lstm_pers_fwd_bf16<1024, 2, 2> spills 5 VGPRs with 8; the spill-free 7 measured SLOWER: 4.68 vs 4.57 us per frame at N = 256)
#endif
#ifndef PERS_PD_BWD2
#define PERS_PD_BWD2 6
#endif

// Bias gradients of a backward launch: every thread holds its share bs[g] of the column sums of dG (its elements over all
// frames) for hidden unit `unit` (0 .. NU-1 inside this workgroup's block at j0).  The workgroup's sums are formed in a FIXED
// order (thread order; the round-5 form added them with LDS atomics, whose order changes from run to run), then either
// STORED into this row group's slab of a.dbp (round 6: run-to-run bit-identical gradients, the default) or added to
// a.db1 / a.db2 with global atomics (the older interface).  The frame loop's LDS is dead here; the launch has >= 84 KB.
template <int NU>
__device__ __forceinline__ void pers_bias_out(const PersArgs& a, char* lds_raw, const float (&bs)[4], int unit, int rb,
                                              int j0, int H) {
  if (!(a.db1 || a.db2 || a.dbp)) return;
  __syncthreads();
  float* val = reinterpret_cast<float*>(lds_raw + 65536);        // [4][256]
  int* un = reinterpret_cast<int*>(lds_raw + 65536 + 4096);      // [256]
  const int tid = threadIdx.x;
#pragma unroll
  for (int g = 0; g < 4; ++g) val[g * 256 + tid] = bs[g];
  un[tid] = unit;
  __syncthreads();
  if (tid < 4 * NU) {
    const int g = tid / NU, u = tid % NU;
    float v = 0.f;
    for (int t = 0; t < 256; ++t) v += (un[t] == u) ? val[g * 256 + t] : 0.f;
    if (a.dbp) {
      a.dbp[((int64_t)rb * 4 + g) * H + j0 + u] = v;
      // the caller always adds DVAE_PERS_BIAS_SLABS row-group slabs: row group 0 clears the ones this launch does not have
      if (rb == 0)
        for (int r2 = a.n_rb; r2 < DVAE_PERS_BIAS_SLABS; ++r2) a.dbp[((int64_t)r2 * 4 + g) * H + j0 + u] = 0.f;
    } else {
      if (a.db1) atomicAdd(a.db1 + g * H + j0 + u, v);
      if (a.db2) atomicAdd(a.db2 + g * H + j0 + u, v);
    }
  }
}

// ======================================================================================================================
// forward:  G = Xproj[t] + h[t-1] W_hh^T ; i,f,o = sigmoid, g = tanh ; c = f c' + i g ; h = o tanh(c)
// ======================================================================================================================
// KL = k-chunks of a wave's W_hh slice kept in LDS instead of registers (MT = 2 at H = 1024: 256 registers of W_hh
// next to 64 accumulators and the fragments in flight spilled; 64 KB of LDS are free)
template <int MT, int KL>
struct FwdLds {
  bf16x8 wl[KL > 0 ? NWV : 1][4][2][KL > 0 ? KL : 1][64];   // [wave][g][u][k - KR][lane]
  f32x4 red[NWV][MT * 8][64];            // partial gate tiles [wave][(mt*4+g)*2+u][lane]
  float stage[16 * MT][6 * 32 + 4];      // per row: activated i,f,g,o, c, h (fp32) of the 32 units -> whole-line stores
                                         // (+4: rows 4 apart — the q groups of a wave — land in different banks)
  __bf16 hx[MT][16][40];                 // h in A-fragment order (32 units + pad per row)
  int dead;
  int loc;                               // pers_loc_check's verdict (frame 1)
};

template <int H, int MT, int KL>
__global__ __launch_bounds__(64 * NWV, 1) void lstm_pers_fwd_bf16(const PersArgs a) {
  constexpr int NCH = H / 32;           // 32-deep k-chunks of h = producers of a row group
  constexpr int KW = NCH / NWV;         // chunks per wave
  constexpr int KR = KW - KL;           // ... of which in registers
  constexpr int NEL = 2 * MT;           // (segment, unit) elements per thread: e = e0 .. e0 + NEL - 1 of ONE (mt, u) tile
  constexpr int PD = (KW * MT > 8) ? PERS_PD_FWD2 : KW;   // chunks of h[t-1] in flight (registers: most hold W_hh)
  typedef float fvec __attribute__((ext_vector_type(NEL)));
  const int T = a.T, N = a.N;
  const int bid = blockIdx.x;
  const int rb = bid % a.n_rb, jb = bid / a.n_rb;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, q = lane >> 4;
  const bool s16 = a.s16 != 0;

  extern __shared__ __attribute__((aligned(16))) char lds_raw[];
  FwdLds<MT, KL>& L = *reinterpret_cast<FwdLds<MT, KL>*>(lds_raw);
  volatile int* dead = &L.dead;
  if (tid == 0) { *dead = 0; L.loc = 0; }
  const unsigned epoch = pers_epoch(a);
  const unsigned long long loc_tag = pers_loc_tag(epoch, pers_xcc());
  bool loc = false;                     // XCD-local hand-off from frame 1 on (pers_loc_*)

  // resident W_hh fragments: gate g, 16-unit tile u, chunk k of this wave's k-quarter
  bf16x8 W[4][2][KR];
#pragma unroll
  for (int g = 0; g < 4; ++g)
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int k = 0; k < KW; ++k) {
        const bf16x8 w = *reinterpret_cast<const bf16x8*>(
            a.wp + ((((int64_t)(g * (H / 16) + 2 * jb + u)) * NCH + wave * KW + k) * 64 + lane) * 16);
        if (k < KR) W[g][u][k < KR ? k : 0] = w;
        else L.wl[wave][g][u][k >= KR ? k - KR : 0][lane] = w;
      }

  // this thread's elements: accumulator-layout positions e0 .. e0+NEL-1 of tile (mt, u): rows mt*16 + q*4 + e, unit u*16 + r
  const int tile = (wave * NEL) >> 2, e0 = (wave * NEL) & 3, emt = tile >> 1, eu = tile & 1;
  const int erow0 = emt * 16 + q * 4 + e0, eunit = eu * 16 + r;
  int el_n[NEL];
  float creg[NEL];
#pragma unroll
  for (int i = 0; i < NEL; ++i) {
    el_n[i] = min(rb * 16 * MT + erow0 + i, N - 1);
    creg[i] = 0.f;
  }
  const int j0 = jb * 32;
  const int64_t H4 = 4 * (int64_t)H;

  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)a.xch, 0, a.xch_bytes, 0x00020000);
  const int slot_bytes = a.n_rb * NCH * MT * 1024;
  const int xld = (rb * NCH + wave * KW) * MT * 1024 + lane * 16;      // this wave's first fragment inside a slot
  const int xst = (rb * NCH + jb) * MT * 1024 + lane * 16;             // where this workgroup publishes
  const unsigned* pflag = a.flags + rb * PERS_FLAG_LD + (lane < NCH ? lane : 0) * PERS_FLAG_STRIDE;
  unsigned* myflag = a.flags + rb * PERS_FLAG_LD + jb * PERS_FLAG_STRIDE;

  auto fetch = [&](int step_, float (&x)[NEL][4]) {
    const int t_ = a.reverse ? (T - 1 - step_) : step_;
    const float* __restrict__ G_ = a.gates + (int64_t)t_ * N * H4 + j0 + eunit;
#pragma unroll
    for (int i = 0; i < NEL; ++i)
#pragma unroll
      for (int g = 0; g < 4; ++g) x[i][g] = G_[(int64_t)el_n[i] * H4 + g * H];
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

    PERS_STAMP(0);
    if (step > 0) {
      if (wave == NWV - 1) {
        if (!poll_ge(pflag, lane < NCH, epoch + (unsigned)step, a.timeout)) {
          pers_give_up(a.err, 1, bid, step, wave);
          *dead = 1;
        } else if (a.local_ok && step == 1) {
          const bool ok = pers_loc_check(pflag, lane < NCH, loc_tag);
          if (lane == 0) L.loc = ok;
        }
      }
      __syncthreads();                                             // barrier A
      if (a.local_ok && step == 1) {
        loc = L.loc != 0;
        if (loc && bid == 0 && tid == 0) atomicAdd(a.err + PERS_LOCAL_WORD, 1u);
      }
      PERS_STAMP(1);
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
        for (int g = 0; g < 4; ++g)
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const bf16x8 w = (k < KR) ? W[g][u][k < KR ? k : 0] : L.wl[wave][g][u][k >= KR ? k - KR : 0][lane];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
              acc[mt][g][u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[k % PD][mt], w, acc[mt][g][u], 0, 0, 0);
          }
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
        for (int u = 0; u < 2; ++u) L.red[wave][(mt * 4 + g) * 2 + u][lane] = acc[mt][g][u];
    PERS_STAMP(2);
    __syncthreads();                                               // barrier B
    PERS_STAMP(3);
    if (*dead) return false;

    {
      fvec gs[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int ti = (emt * 4 + g) * 2 + eu;
        fvec sacc = *reinterpret_cast<const fvec*>(reinterpret_cast<const float*>(&L.red[0][ti][lane]) + e0);
#pragma unroll
        for (int w = 1; w < NWV; ++w)
          sacc += *reinterpret_cast<const fvec*>(reinterpret_cast<const float*>(&L.red[w][ti][lane]) + e0);
        gs[g] = sacc;
      }
#pragma unroll
      for (int i = 0; i < NEL; ++i) {
        const float gi = gate_sigmoid(gs[0][i] + xp[i][0]);
        const float gf = gate_sigmoid(gs[1][i] + xp[i][1]);
        const float gg = gate_tanh(gs[2][i] + xp[i][2]);
        const float go = gate_sigmoid(gs[3][i] + xp[i][3]);
        const float c = gf * creg[i] + gi * gg;
        const float h = go * gate_tanh(c);
        creg[i] = c;
        const int row = erow0 + i;
        L.hx[row >> 4][row & 15][eunit] = (__bf16)h;
        L.stage[row][0 * 32 + eunit] = gi;
        L.stage[row][1 * 32 + eunit] = gf;
        L.stage[row][2 * 32 + eunit] = gg;
        L.stage[row][3 * 32 + eunit] = go;
        L.stage[row][4 * 32 + eunit] = c;
        L.stage[row][5 * 32 + eunit] = h;
      }
    }
    PERS_STAMP(4);
    __syncthreads();                                               // barrier C
    PERS_STAMP(5);
    // lane (r, q) of a wave: units 8q..8q+7 of row r = the 16 bytes of the A fragment AND of the bf16 h_out row
    if (wave == 0) {
      if ((step + 1 < T) && (bid != a.drop_bid)) {
        const int so = (step & 1) * slot_bytes;
        const bool lp = pers_loc_step(loc, step, T);
        if (a.local_ok && step == 0 && lane == 0) pers_loc_announce(myflag, loc_tag);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          const f32x4 v = *reinterpret_cast<const f32x4*>(&L.hx[mt][r][q * 8]);
          PERS_ST(lp, __builtin_bit_cast(u32x4v, v), xrs, xst + mt * 1024, so);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the payload has left (write-through) / reached the L2 (local) before the flag does
        PERS_STAMP(6);
        if (lane == 0) PERS_FLAG(lp, myflag, epoch + (unsigned)(step + 1));
      }
    } else if (wave == 1 && s16) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int n = rb * 16 * MT + mt * 16 + r;
        if (n < N)
          *reinterpret_cast<f32x4*>(reinterpret_cast<__bf16*>(a.h_out) + ((int64_t)t * N + n) * a.ldh + j0 + q * 8) =
              *reinterpret_cast<const f32x4*>(&L.hx[mt][r][q * 8]);
      }
    }
    // the frame's fp32 outputs: 16-byte pieces of whole 128-byte lines, off the hand-off's critical path
    {
      const int npc = s16 ? 5 : 6;                       // i, f, g, o, c (, h when the state is kept in fp32)
      const int pieces = 16 * MT * npc * 8;
      for (int p = tid; p < pieces; p += 64 * NWV) {
        const int row = p / (npc * 8), rem = p - row * (npc * 8), k = rem >> 3, seg = rem & 7;
        const int n = rb * 16 * MT + row;
        if (n < N) {
          const f32x4 v = *reinterpret_cast<const f32x4*>(&L.stage[row][k * 32 + seg * 4]);
          float* dst = k < 4 ? a.gates + ((int64_t)t * N + n) * H4 + k * H + j0 + seg * 4
                     : k == 4 ? a.c_all + ((int64_t)t * N + n) * H + j0 + seg * 4
                              : reinterpret_cast<float*>(a.h_out) + ((int64_t)t * N + n) * a.ldh + j0 + seg * 4;
          *reinterpret_cast<f32x4*>(dst) = v;
        }
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
  if (loc && *dead && wave == 0) {      // gave up with lines in the L2: both slots' pieces and the flag once more, write-through
    const u32x4v z = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int sl = 0; sl < 2; ++sl)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) __builtin_amdgcn_raw_buffer_store_b128(z, xrs, xst + mt * 1024, sl * slot_bytes, 16);
    if (lane == 0) pers_loc_scrub_flag(myflag);
  }
  pers_finish(a);
}

// ======================================================================================================================
// backward:  dH = dHout[t] + dG[t+1] W_hh ; gate derivatives -> dG[t] ; dC carry in registers
// ======================================================================================================================
template <int MT, int KL>
struct BwdLds {
  bf16x8 wl[KL > 0 ? NWV : 1][4][2][KL > 0 ? KL : 1][64];   // [wave][g][u][k - KR][lane]
  f32x4 red[NWV][MT * 2][64];            // partial dH tiles [wave][mt*2+u][lane]
  float stage[16 * MT][4 * 32 + 4];      // fp32 dG (state kept in fp32 only)
  __bf16 gx[4 * MT][16][40];             // dG in A-fragment order [(g, mt)][row][32 units + pad]
  float bsum[4][32];                     // bias gradient of this workgroup's 128 gate columns (summed at the end)
  int dead;
  int loc4[4];                           // pers_loc_check's verdict per wave (frame 1)
};

template <int H, int MT, int KL>
__global__ __launch_bounds__(64 * NWV, 1) void lstm_pers_bwd_bf16(const PersArgs a) {
  constexpr int NCH = H / 32;           // chunks per gate = producers of a row group
  constexpr int KW = NCH / NWV;
  constexpr int KR = KW - KL;
  constexpr int NEL = 2 * MT;
  constexpr int NC = 4 * KW;            // chunks a wave contracts per frame (its unit quarter of all four gates)
  constexpr int PD = (NC * MT > 32) ? PERS_PD_BWD2 : (NC > 8 ? 8 : NC);   // chunks of dG in flight (registers: most hold W_hh)
  typedef float fvec __attribute__((ext_vector_type(NEL)));
  const int T = a.T, N = a.N;
  const int bid = blockIdx.x;
  const int rb = bid % a.n_rb, jb = bid / a.n_rb;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, q = lane >> 4;
  const bool s16 = a.s16 != 0;

  extern __shared__ __attribute__((aligned(16))) char lds_raw[];
  BwdLds<MT, KL>& L = *reinterpret_cast<BwdLds<MT, KL>*>(lds_raw);
  volatile int* dead = &L.dead;
  if (tid == 0) *dead = 0;
  if (tid < 4) L.loc4[tid] = 0;
  const unsigned epoch = pers_epoch(a);
  const unsigned long long loc_tag = pers_loc_tag(epoch, pers_xcc());
  bool loc = false;                     // XCD-local hand-off from frame 1 on (pers_loc_*)
  if (tid < 128) L.bsum[tid >> 5][tid & 31] = 0.f;
  float bs[4] = {0.f, 0.f, 0.f, 0.f};   // this thread's share of the bias gradient: its elements' dG over all frames

  // resident W_hh fragments (rows g*H + k-quarter of this wave, columns = the 32 units of this workgroup)
  bf16x8 W[4][2][KR];
#pragma unroll
  for (int g = 0; g < 4; ++g)
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int k = 0; k < KW; ++k) {
        const bf16x8 w = *reinterpret_cast<const bf16x8*>(
            a.wp + ((((int64_t)((2 * jb + u) * 4 + g)) * NCH + wave * KW + k) * 64 + lane) * 16);
        if (k < KR) W[g][u][k < KR ? k : 0] = w;
        else L.wl[wave][g][u][k >= KR ? k - KR : 0][lane] = w;
      }

  const int tile = (wave * NEL) >> 2, e0 = (wave * NEL) & 3, emt = tile >> 1, eu = tile & 1;
  const int erow0 = emt * 16 + q * 4 + e0, eunit = eu * 16 + r;
  int el_n[NEL];
  bool el_ok[NEL];
  float dcreg[NEL], ccreg[NEL];
#pragma unroll
  for (int i = 0; i < NEL; ++i) {
    el_ok[i] = rb * 16 * MT + erow0 + i < N;
    el_n[i] = min(rb * 16 * MT + erow0 + i, N - 1);
    dcreg[i] = 0.f;
  }
  const int j0 = jb * 32;
  const int64_t H4 = 4 * (int64_t)H;

  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)a.xch, 0, a.xch_bytes, 0x00020000);
  const int slot_bytes = a.n_rb * 4 * NCH * MT * 1024;
  const int xld = (rb * 4 * NCH + wave * KW) * MT * 1024 + lane * 16;     // + g*NCH*MT*1024 per gate
  const int xst = (rb * 4 * NCH + jb) * MT * 1024 + lane * 16;
  const unsigned* pflag = a.flags + rb * PERS_FLAG_LD + (wave * KW + (lane < KW ? lane : 0)) * PERS_FLAG_STRIDE;
  unsigned* myflag = a.flags + rb * PERS_FLAG_LD + jb * PERS_FLAG_STRIDE;

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
      for (int g = 0; g < 4; ++g) o.gt[i][g] = a.gates[((int64_t)t_ * N + el_n[i]) * H4 + g * H + j0 + eunit];
      o.cp[i] = a.c_all[((int64_t)tp_ * N + el_n[i]) * H + j0 + eunit];
      o.dho[i] = a.dh_out[((int64_t)t_ * N + el_n[i]) * a.ldh + j0 + eunit];
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

    PERS_STAMP(0);
    if (step > 0) {
      // every wave waits for the KW producers of ITS k-quarter only (one lane each) and goes on alone — no barrier A
      // (round 6; one polling wave + a barrier: 4.16 -> 3.99 us per frame at H = 1024, N = 128; 3.01 -> 2.87 at H = 512;
      // 6.27 -> 6.20 at N = 256.  The same change made the forward kernels and both fp32x3 16-unit kernels SLOWER by
      // 3-7 %, so only this kernel and lstm_pers_fwd_x3h poll per wave).  LDS is safe without the barrier: `red` of the
      // last frame was read in front of its barrier C; stage / gx are rewritten behind this frame's barrier B only
      if (!poll_ge(pflag, lane < KW, epoch + (unsigned)step, a.timeout)) {
        pers_give_up(a.err, 2, bid, step, wave);
        *dead = 1;
      } else if (a.local_ok && step == 1) {      // this wave's producers; the four verdicts meet behind barrier B
        const bool ok = pers_loc_check(pflag, lane < KW, loc_tag);
        if (lane == 0) L.loc4[wave] = ok;
      }
      PERS_STAMP(1);
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
        for (int u = 0; u < 2; ++u) {
          const bf16x8 w = (k < KR) ? W[g][u][k < KR ? k : 0] : L.wl[wave][g][u][k >= KR ? k - KR : 0][lane];
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
            acc[mt][u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[c % PD][mt], w, acc[mt][u], 0, 0, 0);
        }
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
      for (int u = 0; u < 2; ++u) L.red[wave][mt * 2 + u][lane] = acc[mt][u];
    PERS_STAMP(2);
    __syncthreads();                                               // barrier B
    PERS_STAMP(3);
    if (*dead) return false;
    if (a.local_ok && step == 1) {
      loc = (L.loc4[0] & L.loc4[1] & L.loc4[2] & L.loc4[3]) != 0;
      if (loc && bid == 0 && tid == 0) atomicAdd(a.err + PERS_LOCAL_WORD, 1u);
    }

    {
      const int ti = emt * 2 + eu;
      fvec rec = *reinterpret_cast<const fvec*>(reinterpret_cast<const float*>(&L.red[0][ti][lane]) + e0);
#pragma unroll
      for (int w = 1; w < NWV; ++w)
        rec += *reinterpret_cast<const fvec*>(reinterpret_cast<const float*>(&L.red[w][ti][lane]) + e0);
#pragma unroll
      for (int i = 0; i < NEL; ++i) {
        const float dh = cur.dho[i] + rec[i];
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
        const int row = erow0 + i;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const __bf16 ob = (__bf16)o[g];
          L.gx[g * MT + (row >> 4)][row & 15][eunit] = ob;
          // the bias gradient sums dG as STORED (bf16 when the gate gradients are kept in bf16: oracle/bf16_ref.py)
          if (el_ok[i]) bs[g] += s16 ? (float)ob : o[g];
        }
        if (!s16) {
#pragma unroll
          for (int g = 0; g < 4; ++g) L.stage[row][g * 32 + eunit] = o[g];
        }
      }
    }
    PERS_STAMP(4);
    __syncthreads();                                               // barrier C
    PERS_STAMP(5);
    if (wave == 0) {
      if ((step + 1 < T) && (bid != a.drop_bid)) {
        const int so = (step & 1) * slot_bytes;
        const bool lp = pers_loc_step(loc, step, T);
        if (a.local_ok && step == 0 && lane == 0) pers_loc_announce(myflag, loc_tag);
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(&L.gx[g * MT + mt][r][q * 8]);
            PERS_ST(lp, __builtin_bit_cast(u32x4v, v), xrs, xst + (g * NCH * MT + mt) * 1024, so);
          }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        PERS_STAMP(6);
        if (lane == 0) PERS_FLAG(lp, myflag, epoch + (unsigned)(step + 1));
      }
    } else if (s16) {
      // waves 1..3 archive dG[t] (bf16, whole 64-byte row pieces) for the weight-gradient / dx contractions
      for (int f = wave - 1; f < 4 * MT; f += NWV - 1) {
        const int g = f / MT, mt = f - g * MT;
        const int n = rb * 16 * MT + mt * 16 + r;
        if (n < N)
          *reinterpret_cast<f32x4*>(reinterpret_cast<__bf16*>(a.dgates) + ((int64_t)t * N + n) * H4 + g * H + j0 + q * 8) =
              *reinterpret_cast<const f32x4*>(&L.gx[f][r][q * 8]);
      }
    }
    if (!s16) {
      const int pieces = 16 * MT * 4 * 8;
      for (int p = tid; p < pieces; p += 64 * NWV) {
        const int row = p >> 5, k = (p >> 3) & 3, seg = p & 7;
        const int n = rb * 16 * MT + row;
        if (n < N)
          *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(a.dgates) + ((int64_t)t * N + n) * H4 + k * H + j0 + seg * 4) =
              *reinterpret_cast<const f32x4*>(&L.stage[row][k * 32 + seg * 4]);
      }
    }
    return true;
  };

  // the first frame's cell state, then frame operands one frame ahead (two register sets swapping roles)
  {
    const int t0 = frame_t(0);
#pragma unroll
    for (int i = 0; i < NEL; ++i) ccreg[i] = a.c_all[((int64_t)t0 * N + el_n[i]) * H + j0 + eunit];
  }
  Ops oa, ob;
  fetch(0, oa);
  __syncthreads();
  for (int step = 0; step < T; step += 2) {
    if (!frame(step, oa, ob)) break;
    if (step + 1 < T && !frame(step + 1, ob, oa)) break;
  }
  if (loc && *dead && wave == 0) {      // gave up with lines in the L2: both slots' pieces and the flag once more, write-through
    const u32x4v z = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int sl = 0; sl < 2; ++sl)
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
          __builtin_amdgcn_raw_buffer_store_b128(z, xrs, xst + (g * NCH * MT + mt) * 1024, sl * slot_bytes, 16);
    if (lane == 0) pers_loc_scrub_flag(myflag);
  }
  // db_ih = db_hh = sum over frames and rows of dG (replaces a colsum pass over [T*N, 4H])
  pers_bias_out<32>(a, lds_raw, bs, eunit, rb, j0, H);
  pers_finish(a);
}

// ======================================================================================================================
// fp32x3 forward (the default arithmetic, BASELINE configs[1]/[3]): fp32 RESULTS on the bf16 pipe.  W_hh lives as THREE bf16
// planes (w = w0 + w1 + w2 exactly, dvae_lstm_pack_w_x3), h is handed over as three planes too (the PRODUCER splits each
// value once; 64 consumers read it), six exact partial products per (fragment, column tile) — lstm.hip, PM = 2.
// 3 x the bytes of the bf16 mode per W_hh element, so a workgroup owns 16 hidden units (64 gate columns) x 32 segments:
// planes 0 and 1 of its slice in registers (256 at H = 1024), plane 2 (used by one product of six) mostly in LDS.
// Group = the H/16 workgroups of a row block (64 at H = 1024: one poller lane each).
// ======================================================================================================================
// MT = row tiles of 16 segments per workgroup: 2 (32 rows; H = 1024 at N = 128 fills the chip with 64 x 4 workgroups) or 1 (16
// rows: H = 512 at N = 128 would leave half the CUs idle at 32 rows — 32 x 8 workgroups of 16 rows use all of them)
template <int MT>
struct x3_own { typedef f32x2 type; };
template <>
struct x3_own<1> { typedef float type; };

template <int KW, int K2L, int MT = 2>
struct X3Lds {
  bf16x8 w2[K2L > 0 ? NWV : 1][4][K2L > 0 ? K2L : 1][64];   // plane 2 of chunks KW-K2L .. KW-1: [wave][g][k][lane]
  typename x3_own<MT>::type red[NWV][NWV - 1][4][64];   // partial gate sums FOR wave w's elements FROM the three other waves
  __bf16 hx[3][MT][16][24];              // h planes in A-fragment order [plane][mt][row][16 units + pad]
  int dead;
};

template <int H, int K2L, int MT = 2>
__global__ __launch_bounds__(64 * NWV, 1) void lstm_pers_fwd_x3(const PersArgs a) {
  constexpr int NCH = H / 32;           // 32-deep k-chunks of h
  constexpr int NPR = H / 16;           // producers of a row group
  constexpr int KW = NCH / NWV;         // chunks per wave
  constexpr int K2R = KW - K2L;         // chunks whose plane 2 stays in registers
  constexpr int NEL = MT;               // elements per thread: 16 MT rows x 16 units over 256 threads
  typedef typename x3_own<MT>::type own_t;
  constexpr int RD = PERS_RD_X3;        // (chunk, row tile) units of h[t-1] in flight, three 1-KiB loads per lane each
  const int T = a.T, N = a.N;
  const int bid = blockIdx.x;
  const int rb = bid % a.n_rb, jb = bid / a.n_rb;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, q = lane >> 4;

  extern __shared__ __attribute__((aligned(16))) char lds_raw[];
  X3Lds<KW, K2L, MT>& L = *reinterpret_cast<X3Lds<KW, K2L, MT>*>(lds_raw);
  volatile int* dead = &L.dead;
  if (tid == 0) *dead = 0;
  const unsigned epoch = pers_epoch(a);

  // resident W_hh fragments: [(g*(H/16) + jb)][chunk][plane][lane][8]
  bf16x8 W01[4][KW][2], W2[4][K2R > 0 ? K2R : 1];
#pragma unroll
  for (int g = 0; g < 4; ++g)
#pragma unroll
    for (int k = 0; k < KW; ++k)
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        const bf16x8 w = *reinterpret_cast<const bf16x8*>(
            a.wp + (((((int64_t)(g * (H / 16) + jb)) * NCH + wave * KW + k) * 3 + p) * 64 + lane) * 16);
        if (p < 2) W01[g][k][p < 2 ? p : 0] = w;
        else if (k < K2R) W2[g][k < K2R ? k : 0] = w;
        else L.w2[wave][g][k >= K2R ? k - K2R : 0][lane] = w;
      }

  // this wave owns the elements (row tile emt, accumulator registers e0, e0+1) of every lane: rows emt*16 + q*4 + e0 + {0,1},
  // unit r.  Its accumulators START from those elements' pre-activations (the other positions from zero), so the sum over
  // the four waves' partial tiles IS the gate pre-activation and no operand of the epilogue stays live under the MFMAs.
  // (MT = 1: one element per thread — accumulator register e0 = wave of the one row tile)
  const int emt = MT == 2 ? (wave >> 1) : 0, e0 = MT == 2 ? (wave & 1) * 2 : wave;
  const int erow0 = emt * 16 + q * 4 + e0;
  int el_n[NEL];
  bool el_ok[NEL];
  float creg[NEL];
#pragma unroll
  for (int i = 0; i < NEL; ++i) {
    el_ok[i] = rb * 16 * MT + erow0 + i < N;
    el_n[i] = min(rb * 16 * MT + erow0 + i, N - 1);
    creg[i] = 0.f;
  }
  const int j0 = jb * 16;
  const int64_t H4 = 4 * (int64_t)H;

  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)a.xch, 0, a.xch_bytes, 0x00020000);
  const int slot_bytes = a.n_rb * NCH * MT * 3 * 1024;          // fragment (chunk, mt, plane) = 1 KiB
  const int xld = (rb * NCH + wave * KW) * MT * 3 * 1024 + lane * 16;
  // this workgroup's 16 units are half of chunk jb/2: lanes q' = 2*(jb&1) + {0,1} of its fragments
  const int xst = ((rb * NCH + (jb >> 1)) * MT * 3) * 1024 + ((jb & 1) * 32 + lane) * 16;
  const unsigned* pflag = a.flags + rb * PERS_FLAG_LD_X3 + lane * PERS_FLAG_STRIDE;
  unsigned* myflag = a.flags + rb * PERS_FLAG_LD_X3 + jb * PERS_FLAG_STRIDE;

  auto fetch = [&](int step_, float (&x)[NEL][4]) {
    const int t_ = a.reverse ? (T - 1 - step_) : step_;
    const float* __restrict__ G_ = a.gates + (int64_t)t_ * N * H4 + j0 + r;
#pragma unroll
    for (int i = 0; i < NEL; ++i)
#pragma unroll
      for (int g = 0; g < 4; ++g) x[i][g] = G_[(int64_t)el_n[i] * H4 + g * H];
  };

  float x[NEL][4];                      // pre-activations of the NEXT frame (fetched under this frame's MFMAs)
  fetch(0, x);
  __syncthreads();
  for (int step = 0; step < T; ++step) {
    const int t = a.reverse ? (T - 1 - step) : step;
    f32x4 acc[MT][4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      f32x4 own = f32x4{0.f, 0.f, 0.f, 0.f};
      if constexpr (MT == 2) {
        if (e0) { own[2] = x[0][g]; own[3] = x[1][g]; } else { own[0] = x[0][g]; own[1] = x[1][g]; }
        acc[0][g] = emt ? f32x4{0.f, 0.f, 0.f, 0.f} : own;
        acc[1][g] = emt ? own : f32x4{0.f, 0.f, 0.f, 0.f};
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) own[e] = (e == e0) ? x[0][g] : 0.f;
        acc[0][g] = own;
      }
    }

#ifdef DVAE_DEV
    if (a.dbg) *reinterpret_cast<f32x4*>(a.dbg + (((int64_t)step * gridDim.x + bid) * 256 + tid) * 12 + 4) = f32x4{x[0][0], x[0][1], x[0][2], x[0][3]};
#endif
    PERS_STAMP(0);
    if (step > 0) {
      if (wave == NWV - 1 && !poll_ge(pflag, lane < NPR, epoch + (unsigned)step, a.timeout)) {
        pers_give_up(a.err, 1, bid, step, wave);
        *dead = 1;
      }
      __syncthreads();                                             // barrier A
      PERS_STAMP(1);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#ifdef DVAE_DEV
      const int so = ((step - 1) % a.nslot) * slot_bytes;
      u32x4v dbg_ck = {0u, 0u, 0u, 0u};
#else
      const int so = ((step - 1) & 1) * slot_bytes;
#endif
      // unit u = (chunk k, row tile mt) = three 1-KiB fragments (the planes of h); RD units in flight
      bf16x8 av[RD][3];
      auto load = [&](int u) {
#pragma unroll
        for (int p = 0; p < 3; ++p)
          av[u % RD][p] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(xrs, xld + (u * 3 + p) * 1024, so, 16));
      };
#pragma unroll
      for (int u = 0; u < RD; ++u) load(u);
      fetch(min(step + 1, T - 1), x);
      __builtin_amdgcn_sched_barrier(0);
      bf16x8 w2[4];                           // plane 2 of chunk k: read from LDS ONCE per chunk, used by both row tiles
#pragma unroll
      for (int u = 0; u < KW * MT; ++u) {
        const int k = u / MT, mt = u % MT;
        if (mt == 0) {
#pragma unroll
          for (int g = 0; g < 4; ++g)
            w2[g] = (k < K2R) ? W2[g][k < K2R ? k : 0] : L.w2[wave][g][k >= K2R ? k - K2R : 0][lane];
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          // the six partial products of weight >= 2^-16: (h plane, W plane)
          acc[mt][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[u % RD][0], W01[g][k][0], acc[mt][g], 0, 0, 0);
          acc[mt][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[u % RD][0], W01[g][k][1], acc[mt][g], 0, 0, 0);
          acc[mt][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[u % RD][1], W01[g][k][0], acc[mt][g], 0, 0, 0);
          acc[mt][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[u % RD][1], W01[g][k][1], acc[mt][g], 0, 0, 0);
          acc[mt][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[u % RD][0], w2[g], acc[mt][g], 0, 0, 0);
          acc[mt][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[u % RD][2], W01[g][k][0], acc[mt][g], 0, 0, 0);
        }
#ifdef DVAE_DEV
        if (a.dbg) {
#pragma unroll
          for (int p = 0; p < 3; ++p) {
            const u32x4v raw = __builtin_bit_cast(u32x4v, av[u % RD][p]);
            dbg_ck ^= raw;
            if (a.dbg_frag && jb == a.dbg_jb)
              *reinterpret_cast<u32x4v*>(a.dbg_frag + (((((int64_t)step * a.n_rb + rb) * NWV + wave) * (KW * MT * 3) + u * 3 + p) * 64 + lane) * 4) = raw;
          }
        }
#endif
        if (u + RD < KW * MT) {
          __builtin_amdgcn_sched_barrier(0);
          load(u + RD);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
#ifdef DVAE_DEV
      if (a.dbg) *reinterpret_cast<u32x4v*>(a.dbg + (((int64_t)step * gridDim.x + bid) * 256 + tid) * 12) = dbg_ck;
#endif
    } else {
      fetch(min(step + 1, T - 1), x);
    }
    // every wave hands the other three owners their part of its partial tiles and keeps its own in registers
#pragma unroll
    for (int o = 0; o < NWV; ++o) {
      if (o == wave) continue;
      const int slot = (wave - o - 1) & (NWV - 1);               // 0..2
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        if constexpr (MT == 2) {
          const f32x4 v = acc[o >> 1][g];
          L.red[o][slot][g][lane] = (o & 1) ? f32x2{v[2], v[3]} : f32x2{v[0], v[1]};
        } else {
          const f32x4 v = acc[0][g];
          L.red[o][slot][g][lane] = o == 0 ? v[0] : o == 1 ? v[1] : o == 2 ? v[2] : v[3];
        }
      }
    }
    PERS_STAMP(2);
    __syncthreads();                                               // barrier B
    PERS_STAMP(3);
    if (*dead) break;

    float go_[NEL][4], co_[NEL], ho_[NEL];
    {
      float gs[4][NEL];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        own_t sacc;
        if constexpr (MT == 2) {
          f32x4 own = acc[0][g];
          if (emt) own = acc[1][g];
          sacc = e0 ? f32x2{own[2], own[3]} : f32x2{own[0], own[1]};
        } else {
          const f32x4 own = acc[0][g];
          sacc = e0 == 0 ? own[0] : e0 == 1 ? own[1] : e0 == 2 ? own[2] : own[3];
        }
#pragma unroll
        for (int sl = 0; sl < NWV - 1; ++sl) sacc += L.red[wave][sl][g][lane];
        if constexpr (MT == 2) { gs[g][0] = sacc[0]; gs[g][1] = sacc[1]; } else { gs[g][0] = sacc; }
      }
#ifdef DVAE_DEV
      if (a.dbg) *reinterpret_cast<f32x4*>(a.dbg + (((int64_t)step * gridDim.x + bid) * 256 + tid) * 12 + 8) = f32x4{gs[0][0], gs[1][0], gs[2][0], gs[3][0]};
#endif
#pragma unroll
      for (int i = 0; i < NEL; ++i) {
        const float gi = gate_sigmoid(gs[0][i]);
        const float gf = gate_sigmoid(gs[1][i]);
        const float gg = gate_tanh(gs[2][i]);
        const float go = gate_sigmoid(gs[3][i]);
        const float c = gf * creg[i] + gi * gg;
        const float h = go * gate_tanh(c);
        creg[i] = c;
        const int row = erow0 + i;
        // h = h0 + h1 + h2 exactly (two round-to-nearest splits; the last residual is exact in bf16)
        const __bf16 h0 = (__bf16)h;
        const float r1 = h - (float)h0;
        const __bf16 h1 = (__bf16)r1;
        const __bf16 h2 = (__bf16)(r1 - (float)h1);
        L.hx[0][row >> 4][row & 15][r] = h0;
        L.hx[1][row >> 4][row & 15][r] = h1;
        L.hx[2][row >> 4][row & 15][r] = h2;
        go_[i][0] = gi; go_[i][1] = gf; go_[i][2] = gg; go_[i][3] = go;
        co_[i] = c; ho_[i] = h;
      }
    }
    PERS_STAMP(4);
    __syncthreads();                                               // barrier C
    PERS_STAMP(5);
    if (wave == 0) {
      if ((step + 1 < T) && (bid != a.drop_bid)) {
#ifdef DVAE_DEV
        const int so = (step % a.nslot) * slot_bytes;
#else
        const int so = (step & 1) * slot_bytes;
#endif
        if (lane < 32) {                // lane (r, q' in {0,1}): units 8q'..8q'+7 of row r
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int p = 0; p < 3; ++p) {
              const f32x4 v = *reinterpret_cast<const f32x4*>(&L.hx[p][mt][r][(q & 1) * 8]);
              __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4v, v), xrs, xst + (mt * 3 + p) * 1024, so, 16);
            }
        }
        PERS_DRAIN();
        PERS_STAMP(6);
        if (lane == 0) __hip_atomic_store(myflag, epoch + (unsigned)(step + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    // the frame's outputs (activated gates, c, h), after the hand-off has left
#pragma unroll
    for (int i = 0; i < NEL; ++i) {
      if (el_ok[i]) {
        float* gp = a.gates + ((int64_t)t * N + el_n[i]) * H4 + j0 + r;
        gp[0] = go_[i][0];
        gp[H] = go_[i][1];
        gp[2 * H] = go_[i][2];
        gp[3 * H] = go_[i][3];
        a.c_all[((int64_t)t * N + el_n[i]) * H + j0 + r] = co_[i];
        reinterpret_cast<float*>(a.h_out)[((int64_t)t * N + el_n[i]) * a.ldh + j0 + r] = ho_[i];
      }
    }
  }
  pers_finish(a);
}

// ======================================================================================================================
// fp32x3 forward, H = 512 at N <= 128: EIGHT hidden units x 32 rows per workgroup (round 6).
// The 16-unit kernel above fills 32 x 4 = 128 of the 256 CUs there and each of them spends 192 MFMAs per wave and frame;
// its 16-ROW form (MT = 1) fills the chip but is the one geometry whose hand-off has failed on some chips (see
// dvae_pers_launch).  This form halves the COLUMNS instead: 64 x 4 workgroups, 96 MFMAs per wave and frame, and the row
// groups, flags (64 producers per row group, one poller lane each) and ring (32-row fragments, two slots) are exactly
// those of the H = 1024 kernel, which has never failed.
//   column tile tl of the 16x16x32 MFMA = TWO gates x 8 units: columns 0..7 gate 2tl, columns 8..15 gate 2tl+1 (tl 0: i, f;
//   tl 1: g, o) — a lane permutation of the 16-unit fragment pack (block jb >> 1, lanes q*16 + (jb & 1)*8 + (c & 7)), so the
//   pack of repack.hip serves both kernels;  all three planes of the slice stay in registers (96 VGPRs);
//   epilogue: ONE element (row tid / 8, unit tid % 8) per thread; every wave leaves its whole partial tiles in LDS and the
//   owner adds them in wave order to the pre-activation: one summation order, run-to-run bit-identical;
//   hand-off: the workgroup's 8 units are a QUARTER of chunk jb / 4 (lanes (jb & 3)*16 + row of its fragments); wave w waits
//   for the 16 producers of its own four chunks only.
// Measured (N = 128, T = 128, scripts/lstm_rec_bench.py): 4.43 -> 3.70 us per frame.  In-kernel timeline of a frame (3.52 us,
// scripts/lstm_pers_timeline.py): own flag -> all awaited flags seen 0.9, fragment loads (96 KiB per CU, sc1) + 96 MFMAs
// 1.4, barrier B 0.1-0.3, gates 0.4, barrier C 0.04, payload + write-through acknowledgement 0.5: about 2.3 us of it are
// three fabric round trips (flag, first fragment, acknowledgement) that no tiling shortens.
// ======================================================================================================================
struct X3HLds {
  f32x4 red[NWV][2][2][64];             // every wave's partial tiles [wave][row tile][gate pair][lane]
  __bf16 hx[3][2][16][8];               // h planes in A-fragment order [plane][row tile][row][8 units]: 16 B per row
  int dead;
};

template <int H>
__global__ __launch_bounds__(64 * NWV, 1) void lstm_pers_fwd_x3h(const PersArgs a) {
  constexpr int NCH = H / 32;           // 32-deep k-chunks of h
  constexpr int NPR = H / 8;            // producers of a row group
  constexpr int KW = NCH / NWV;         // chunks per wave
  constexpr int MT = 2;
  constexpr int RD = PERS_RD_X3H;
  static_assert(NPR <= 64, "one poller lane per producer");
  const int T = a.T, N = a.N;
  const int bid = blockIdx.x;
  const int rb = bid % a.n_rb, jb = bid / a.n_rb;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, q = lane >> 4;

  extern __shared__ __attribute__((aligned(16))) char lds_raw[];
  X3HLds& L = *reinterpret_cast<X3HLds*>(lds_raw);
  volatile int* dead = &L.dead;
  if (tid == 0) *dead = 0;
  const unsigned epoch = pers_epoch(a);

  // resident W_hh fragments of this wave's k-quarter: [gate pair][chunk][plane]
  bf16x8 W[2][KW][3];
#pragma unroll
  for (int tl = 0; tl < 2; ++tl)
#pragma unroll
    for (int k = 0; k < KW; ++k)
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        const int g = tl * 2 + (r >> 3);
        const int src_lane = q * 16 + (jb & 1) * 8 + (r & 7);
        W[tl][k][p] = *reinterpret_cast<const bf16x8*>(
            a.wp + (((((int64_t)(g * (H / 16) + (jb >> 1))) * NCH + wave * KW + k) * 3 + p) * 64 + src_lane) * 16);
      }

  // this thread's element: row erow of the workgroup's 32, hidden unit j0 + eunit
  const int erow = tid >> 3, eunit = tid & 7;
  const bool el_ok = rb * 32 + erow < N;
  const int el_n = min(rb * 32 + erow, N - 1);
  const int emt = erow >> 4;
  const int esrc = ((erow & 15) >> 2) * 16 + eunit, ee = erow & 3;      // its accumulator: lane esrc (+ 8: second gate), register ee
  float creg = 0.f;
  const int j0 = jb * 8;
  const int64_t H4 = 4 * (int64_t)H;

  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)a.xch, 0, a.xch_bytes, 0x00020000);
  const int slot_bytes = a.n_rb * NCH * MT * 3 * 1024;          // fragment (chunk, mt, plane) = 1 KiB
  const int xld = (rb * NCH + wave * KW) * MT * 3 * 1024 + lane * 16;
  const int xst = ((rb * NCH + (jb >> 2)) * MT * 3 + (lane >> 4) * 3) * 1024 + ((jb & 3) * 16 + r) * 16;   // lanes 0..31: (row tile, row)
  const unsigned* pflag = a.flags + rb * PERS_FLAG_LD_X3 + (wave * 4 * KW + (lane & (4 * KW - 1))) * PERS_FLAG_STRIDE;
  unsigned* myflag = a.flags + rb * PERS_FLAG_LD_X3 + jb * PERS_FLAG_STRIDE;

  auto fetch = [&](int step_, float (&x)[4]) {
    const int t_ = a.reverse ? (T - 1 - step_) : step_;
    const float* __restrict__ G_ = a.gates + ((int64_t)t_ * N + el_n) * H4 + j0 + eunit;
#pragma unroll
    for (int g = 0; g < 4; ++g) x[g] = G_[g * H];
  };

  float xn[4];                          // pre-activations of the NEXT frame (fetched under this frame's MFMAs)
  fetch(0, xn);
  __syncthreads();
  for (int step = 0; step < T; ++step) {
    const int t = a.reverse ? (T - 1 - step) : step;
    float x[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) x[g] = xn[g];
    f32x4 acc[MT][2];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int tl = 0; tl < 2; ++tl) acc[mt][tl] = f32x4{0.f, 0.f, 0.f, 0.f};

    PERS_STAMP(0);
    if (step > 0) {
      // every wave polls the 4 * KW producers of ITS k-quarter (one lane each) and goes on alone — no barrier A (one polling
      // wave + a barrier, the 16-unit kernel's form: 3.90 us per frame instead of 3.70).  LDS is safe without it: `red` of the
      // last frame was read in front of its barrier C, `hx` is rewritten behind this frame's barrier B only
      if (!poll_ge(pflag, lane < 4 * KW, epoch + (unsigned)step, a.timeout)) {
        pers_give_up(a.err, 1, bid, step, wave);
        *dead = 1;
      }
      PERS_STAMP(1);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      const int so = ((step - 1) & 1) * slot_bytes;
      // unit u = (chunk k, row tile mt) = three 1-KiB fragments (the planes of h); RD units in flight
      bf16x8 av[RD][3];
      auto load = [&](int u) {
#pragma unroll
        for (int p = 0; p < 3; ++p)
          av[u % RD][p] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(xrs, xld + (u * 3 + p) * 1024, so, 16));
      };
#pragma unroll
      for (int u = 0; u < RD; ++u) load(u);
      fetch(min(step + 1, T - 1), xn);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < KW * MT; ++u) {
        const int k = u / MT, mt = u % MT;
#pragma unroll
        for (int tl = 0; tl < 2; ++tl) {
          // the six partial products of weight >= 2^-16: (h plane, W plane)
          acc[mt][tl] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[u % RD][0], W[tl][k][0], acc[mt][tl], 0, 0, 0);
          acc[mt][tl] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[u % RD][0], W[tl][k][1], acc[mt][tl], 0, 0, 0);
          acc[mt][tl] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[u % RD][1], W[tl][k][0], acc[mt][tl], 0, 0, 0);
          acc[mt][tl] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[u % RD][1], W[tl][k][1], acc[mt][tl], 0, 0, 0);
          acc[mt][tl] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[u % RD][0], W[tl][k][2], acc[mt][tl], 0, 0, 0);
          acc[mt][tl] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[u % RD][2], W[tl][k][0], acc[mt][tl], 0, 0, 0);
        }
        if (u + RD < KW * MT) {
          __builtin_amdgcn_sched_barrier(0);
          load(u + RD);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    } else {
      fetch(min(step + 1, T - 1), xn);
    }
    // every wave leaves its partial tiles in LDS (16-byte stores)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int tl = 0; tl < 2; ++tl) L.red[wave][mt][tl][lane] = acc[mt][tl];
    PERS_STAMP(2);
    __syncthreads();                                               // barrier B
    PERS_STAMP(3);
    if (*dead) break;

    float gs[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      float s = x[g];
#pragma unroll
      for (int w = 0; w < NWV; ++w)      // wave order: ONE summation order
        s += reinterpret_cast<const float*>(&L.red[w][emt][g >> 1][esrc + (g & 1) * 8])[ee];
      gs[g] = s;
    }
    const float gi = gate_sigmoid(gs[0]);
    const float gf = gate_sigmoid(gs[1]);
    const float gg = gate_tanh(gs[2]);
    const float go = gate_sigmoid(gs[3]);
    const float c = gf * creg + gi * gg;
    const float h = go * gate_tanh(c);
    creg = c;
    {
      // h = h0 + h1 + h2 exactly (two round-to-nearest splits; the last residual is exact in bf16)
      const __bf16 h0 = (__bf16)h;
      const float r1 = h - (float)h0;
      const __bf16 h1 = (__bf16)r1;
      const __bf16 h2 = (__bf16)(r1 - (float)h1);
      L.hx[0][emt][erow & 15][eunit] = h0;
      L.hx[1][emt][erow & 15][eunit] = h1;
      L.hx[2][emt][erow & 15][eunit] = h2;
    }
    PERS_STAMP(4);
    __syncthreads();                                               // barrier C
    PERS_STAMP(5);
    if (wave == 0) {
      if ((step + 1 < T) && (bid != a.drop_bid)) {
        const int so = (step & 1) * slot_bytes;
        if (lane < 32) {                // lane (row tile, row r): its 8 units of the three planes
#pragma unroll
          for (int p = 0; p < 3; ++p) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(&L.hx[p][lane >> 4][r][0]);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4v, v), xrs, xst + p * 1024, so, 16);
          }
        }
        PERS_DRAIN();
        PERS_STAMP(6);
        if (lane == 0) __hip_atomic_store(myflag, epoch + (unsigned)(step + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    // the frame's outputs (activated gates, c, h), after the hand-off has left
    if (el_ok) {
      float* gp = a.gates + ((int64_t)t * N + el_n) * H4 + j0 + eunit;
      gp[0] = gi;
      gp[H] = gf;
      gp[2 * H] = gg;
      gp[3 * H] = go;
      a.c_all[((int64_t)t * N + el_n) * H + j0 + eunit] = c;
      reinterpret_cast<float*>(a.h_out)[((int64_t)t * N + el_n) * a.ldh + j0 + eunit] = h;
    }
  }
  pers_finish(a);
}

#ifdef DVAE_DEV
// ======================================================================================================================
// SENTINEL hand-off (round 5; DEV BUILD ONLY — measured slower than the flag protocol, kept as the reproducer of the 16-row
// forward form's stale reads): the payload is its own flag.  Every ring word that is not yet this frame's data holds a
// POISON no payload can contain (0xFFFFFFFF: two bf16 NaNs / an fp32 NaN with all mantissa bits set); a consumer loads
// the fragments it needs straight into MFMA operand registers as before and looks at them: poison in any dword = not there
// yet, load again.  No flag, no drain in front of it, no poll, no barrier between poll and loads — and a wave may ask for
// the next frame's fragments BEFORE anybody has said they exist (speculative prefetch under the epilogue: for the slowest
// workgroup of a row group, the one that sets the frame time, nearly everything is already there).
//   ring   : FOUR slots; frame s lives in slot s & 3.
//   publish: (wave 0, after barrier C of step s)  s_waitcnt vmcnt(0)  ->  data(s) into slot s & 3 (write-through, no drain)
//            ->  poison into its own piece of slot (s+2) & 3.
//   why that is enough: slot (s+2)&3 held frame s-2, whose last readers finished before they published frame s-1, and this
//   workgroup has checked ALL of frame s-1 during step s: nobody reads those words any more.  A consumer asks for
//   frame s+2 of this producer only after it has SEEN the producer's data(s+1), which was issued behind the vmcnt(0) that
//   waited for the poison stores of step s: the poison is in memory before anybody can look for frame s+2 there, so a
//   consumer sees poison or frame s+2, never frame s-2.  Slots 0 and 1 are poisoned by a small launch in front (every
//   other slot by the producers themselves: step 0 poisons slot 2, step 1 slot 3, step 2 slot 0 ...).
//   check  : every dword of every 16-byte piece (a torn 16-byte store would show as poison in some dword).
//   waiting: the first lane that still sees poison re-reads ITS three pieces alone (one lane: no bandwidth) until they carry
//            data, then the wave loads the unit again; bounded like every wait here.
// The unit order of the wave that owns this workgroup's own chunk is rotated so that the own chunk comes LAST: it is never
// part of the speculative prefetch (which runs before the workgroup has published).
// ======================================================================================================================
constexpr unsigned PERS_POISON = 0xFFFFFFFFu;
constexpr int PERS_NSLOT_S = 4;

__device__ __forceinline__ unsigned umax4(const u32x4v& v) {
  const unsigned a = v[0] > v[1] ? v[0] : v[1], b = v[2] > v[3] ? v[2] : v[3];
  return a > b ? a : b;
}

// wave-level: (re)load the NP pieces of one unit until none of them holds poison; `ld(p)` loads piece p for every active lane
template <int NP, class LD>
__device__ __forceinline__ bool sent_wait(u32x4v (&v)[NP], LD ld, int lane, unsigned timeout) {
  auto poisoned = [&]() {
    unsigned m = umax4(v[0]);
#pragma unroll
    for (int p = 1; p < NP; ++p) {
      const unsigned x = umax4(v[p]);
      m = x > m ? x : m;
    }
    return m == PERS_POISON;
  };
  bool bad = poisoned();
  if (__builtin_expect(!__any(bad), 1)) return true;
  const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
  for (;;) {
    const uint64_t mask = __ballot(bad);
    if (!mask) return true;
    const int first = __builtin_ctzll(mask);
    bool gave_up = false;
    if (lane == first) {
      for (unsigned it = 1;; ++it) {
        __builtin_amdgcn_s_sleep(2);
#pragma unroll
        for (int p = 0; p < NP; ++p) v[p] = ld(p);
        if (!poisoned()) break;
        if ((it & 15) == 0 && (__builtin_amdgcn_s_memrealtime() - t0) > timeout) { gave_up = true; break; }
      }
    }
    if (__any(gave_up)) return false;
#pragma unroll
    for (int p = 0; p < NP; ++p) v[p] = ld(p);
    bad = poisoned();
  }
}

template <int H, int K2L, int MT = 2>
__global__ __launch_bounds__(64 * NWV, 1) void lstm_pers_fwd_x3s(const PersArgs a) {
  constexpr int NCH = H / 32;           // 32-deep k-chunks of h
  constexpr int KW = NCH / NWV;         // chunks per wave
  constexpr int K2R = KW - K2L;         // chunks whose plane 2 stays in registers
  constexpr int NEL = MT;
  typedef typename x3_own<MT>::type own_t;
  constexpr int NU = KW * MT;           // (chunk, row tile) units per wave and frame
  constexpr int RD = PERS_RD_X3 < NU ? PERS_RD_X3 : NU;
  constexpr int NPRE = RD < NU - MT ? RD : NU - MT;    // units asked for speculatively (never the last chunk: see above)
  static_assert(NPRE >= 1, "speculative prefetch needs two chunks per wave");
  const int T = a.T, N = a.N;
  const int bid = blockIdx.x;
  const int rb = bid % a.n_rb, jb = bid / a.n_rb;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, q = lane >> 4;

  extern __shared__ __attribute__((aligned(16))) char lds_raw[];
  X3Lds<KW, K2L, MT>& L = *reinterpret_cast<X3Lds<KW, K2L, MT>*>(lds_raw);
  volatile int* dead = &L.dead;
  if (tid == 0) *dead = 0;

  // position kk of this wave's order is chunk wave*KW + (kk + koff) % KW: the workgroup's own chunk (jb >> 1) comes last
  const int c_own = jb >> 1;
  const int koff = (c_own / KW == wave) ? (c_own % KW + 1) % KW : 0;
  auto chunk_of = [&](int kk) { return wave * KW + (kk + koff) % KW; };

  bf16x8 W01[4][KW][2], W2[4][K2R > 0 ? K2R : 1];
#pragma unroll
  for (int g = 0; g < 4; ++g)
#pragma unroll
    for (int k = 0; k < KW; ++k)
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        const bf16x8 w = *reinterpret_cast<const bf16x8*>(
            a.wp + (((((int64_t)(g * (H / 16) + jb)) * NCH + chunk_of(k)) * 3 + p) * 64 + lane) * 16);
        if (p < 2) W01[g][k][p < 2 ? p : 0] = w;
        else if (k < K2R) W2[g][k < K2R ? k : 0] = w;
        else L.w2[wave][g][k >= K2R ? k - K2R : 0][lane] = w;
      }

  const int emt = MT == 2 ? (wave >> 1) : 0, e0 = MT == 2 ? (wave & 1) * 2 : wave;
  const int erow0 = emt * 16 + q * 4 + e0;
  int el_n[NEL];
  bool el_ok[NEL];
  float creg[NEL];
#pragma unroll
  for (int i = 0; i < NEL; ++i) {
    el_ok[i] = rb * 16 * MT + erow0 + i < N;
    el_n[i] = min(rb * 16 * MT + erow0 + i, N - 1);
    creg[i] = 0.f;
  }
  const int j0 = jb * 16;
  const int64_t H4 = 4 * (int64_t)H;

  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)a.xch, 0, a.xch_bytes, 0x00020000);
  const int slot_bytes = a.n_rb * NCH * MT * 3 * 1024;          // fragment (chunk, mt, plane) = 1 KiB
  const int xrow = rb * NCH * MT * 3 * 1024 + lane * 16;        // + (chunk * MT + mt) * 3072 + plane * 1024
  int xoff[KW];                                                 // per position kk: byte offset of its chunk inside a slot
#pragma unroll
  for (int k = 0; k < KW; ++k) xoff[k] = chunk_of(k) * MT * 3 * 1024;
  const int xst = ((rb * NCH + (jb >> 1)) * MT * 3) * 1024 + ((jb & 1) * 32 + lane) * 16;

  auto fetch = [&](int step_, float (&x)[NEL][4]) {
    const int t_ = a.reverse ? (T - 1 - step_) : step_;
    const float* __restrict__ G_ = a.gates + (int64_t)t_ * N * H4 + j0 + r;
#pragma unroll
    for (int i = 0; i < NEL; ++i)
#pragma unroll
      for (int g = 0; g < 4; ++g) x[i][g] = G_[(int64_t)el_n[i] * H4 + g * H];
  };

  u32x4v av[RD][3];                     // units in flight: live ACROSS frames (the speculative prefetch)
  // unit u = (position kk = u / MT, row tile mt = u % MT) of the frame in slot `so`
  auto load = [&](int u, int so) __attribute__((always_inline)) {
    const int kk = u / MT, mt = u % MT;
#pragma unroll
    for (int p = 0; p < 3; ++p)
      av[u % RD][p] = __builtin_amdgcn_raw_buffer_load_b128(xrs, xrow, so + xoff[kk] + (mt * 3 + p) * 1024, 16);   // ONE address register
  };

#ifdef DVAE_DEV
  // dev build: nslot > 4 = one ring slot per frame, all poisoned in front of the launch, none re-poisoned here (no address
  // is written twice in a launch: scripts/x3_fwd16_diag2.py)
  const bool per_frame = a.nslot > PERS_NSLOT_S;
  auto slot_of = [&](int s_) { return (per_frame ? s_ % a.nslot : (s_ & (PERS_NSLOT_S - 1))) * slot_bytes; };
#else
  constexpr bool per_frame = false;
  auto slot_of = [&](int s_) { return (s_ & (PERS_NSLOT_S - 1)) * slot_bytes; };
#endif
  float x[NEL][4];
  fetch(0, x);
  __syncthreads();
  for (int step = 0; step < T; ++step) {
    const int t = a.reverse ? (T - 1 - step) : step;
    f32x4 acc[MT][4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      f32x4 own = f32x4{0.f, 0.f, 0.f, 0.f};
      if constexpr (MT == 2) {
        if (e0) { own[2] = x[0][g]; own[3] = x[1][g]; } else { own[0] = x[0][g]; own[1] = x[1][g]; }
        acc[0][g] = emt ? f32x4{0.f, 0.f, 0.f, 0.f} : own;
        acc[1][g] = emt ? own : f32x4{0.f, 0.f, 0.f, 0.f};
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) own[e] = (e == e0) ? x[0][g] : 0.f;
        acc[0][g] = own;
      }
    }

    PERS_STAMP(0);
    if (step > 0) {
      const int so = slot_of(step - 1);
      // units 0 .. NPRE-1 were asked for at the end of the previous step
#pragma unroll
      for (int u = NPRE; u < RD; ++u) load(u, so);
      fetch(min(step + 1, T - 1), x);
      __builtin_amdgcn_sched_barrier(0);
      bf16x8 w2[4];
      bool ok = true;
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        const int k = u / MT, mt = u % MT;
        if (mt == 0) {
#pragma unroll
          for (int g = 0; g < 4; ++g)
            w2[g] = (k < K2R) ? W2[g][k < K2R ? k : 0] : L.w2[wave][g][k >= K2R ? k - K2R : 0][lane];
        }
        ok = ok && sent_wait<3>(av[u % RD], [&](int p) __attribute__((always_inline)) {
               return __builtin_amdgcn_raw_buffer_load_b128(xrs, xrow, so + xoff[k] + (mt * 3 + p) * 1024, 16); }, lane, a.timeout);
#ifdef DVAE_DEV
        if (a.dbg_frag && jb == a.dbg_jb) {
#pragma unroll
          for (int p = 0; p < 3; ++p)      // (in the wave's OWN unit order: position kk, not chunk)
            *reinterpret_cast<u32x4v*>(a.dbg_frag + (((((int64_t)step * a.n_rb + rb) * NWV + wave) * (NU * 3) + u * 3 + p) * 64 + lane) * 4) = av[u % RD][p];
        }
#endif
        const bf16x8 h0 = __builtin_bit_cast(bf16x8, av[u % RD][0]), h1 = __builtin_bit_cast(bf16x8, av[u % RD][1]),
                     h2 = __builtin_bit_cast(bf16x8, av[u % RD][2]);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          acc[mt][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(h0, W01[g][k][0], acc[mt][g], 0, 0, 0);
          acc[mt][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(h0, W01[g][k][1], acc[mt][g], 0, 0, 0);
          acc[mt][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(h1, W01[g][k][0], acc[mt][g], 0, 0, 0);
          acc[mt][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(h1, W01[g][k][1], acc[mt][g], 0, 0, 0);
          acc[mt][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(h0, w2[g], acc[mt][g], 0, 0, 0);
          acc[mt][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(h2, W01[g][k][0], acc[mt][g], 0, 0, 0);
        }
        if (u + RD < NU) {
          __builtin_amdgcn_sched_barrier(0);
          load(u + RD, so);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      if (!ok) {
        pers_give_up(a.err, 1, bid, step, wave);
        *dead = 1;
      }
    } else {
      fetch(min(step + 1, T - 1), x);
    }
    // speculative prefetch of the next frame's first units (waves 1..3 here; wave 0 behind its publish below, which
    // must not wait for these loads)
    if (wave != 0 && step + 1 < T) {
#pragma unroll
      for (int u = 0; u < NPRE; ++u) load(u, slot_of(step));
    }
#pragma unroll
    for (int o = 0; o < NWV; ++o) {
      if (o == wave) continue;
      const int slot = (wave - o - 1) & (NWV - 1);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        if constexpr (MT == 2) {
          const f32x4 v = acc[o >> 1][g];
          L.red[o][slot][g][lane] = (o & 1) ? f32x2{v[2], v[3]} : f32x2{v[0], v[1]};
        } else {
          const f32x4 v = acc[0][g];
          L.red[o][slot][g][lane] = o == 0 ? v[0] : o == 1 ? v[1] : o == 2 ? v[2] : v[3];
        }
      }
    }
    PERS_STAMP(2);
    __syncthreads();                                               // barrier B
    PERS_STAMP(3);
    if (*dead) break;

    float go_[NEL][4], co_[NEL], ho_[NEL];
    {
      float gs[4][NEL];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        own_t sacc;
        if constexpr (MT == 2) {
          f32x4 own = acc[0][g];
          if (emt) own = acc[1][g];
          sacc = e0 ? f32x2{own[2], own[3]} : f32x2{own[0], own[1]};
        } else {
          const f32x4 own = acc[0][g];
          sacc = e0 == 0 ? own[0] : e0 == 1 ? own[1] : e0 == 2 ? own[2] : own[3];
        }
#pragma unroll
        for (int sl = 0; sl < NWV - 1; ++sl) sacc += L.red[wave][sl][g][lane];
        if constexpr (MT == 2) { gs[g][0] = sacc[0]; gs[g][1] = sacc[1]; } else { gs[g][0] = sacc; }
      }
#pragma unroll
      for (int i = 0; i < NEL; ++i) {
        const float gi = gate_sigmoid(gs[0][i]);
        const float gf = gate_sigmoid(gs[1][i]);
        const float gg = gate_tanh(gs[2][i]);
        const float go = gate_sigmoid(gs[3][i]);
        const float c = gf * creg[i] + gi * gg;
        const float h = go * gate_tanh(c);
        creg[i] = c;
        const int row = erow0 + i;
        const __bf16 h0 = (__bf16)h;
        const float r1 = h - (float)h0;
        const __bf16 h1 = (__bf16)r1;
        const __bf16 h2 = (__bf16)(r1 - (float)h1);
        L.hx[0][row >> 4][row & 15][r] = h0;
        L.hx[1][row >> 4][row & 15][r] = h1;
        L.hx[2][row >> 4][row & 15][r] = h2;
        go_[i][0] = gi; go_[i][1] = gf; go_[i][2] = gg; go_[i][3] = go;
        co_[i] = c; ho_[i] = h;
      }
    }
    PERS_STAMP(4);
    __syncthreads();                                               // barrier C
    PERS_STAMP(5);
    if (wave == 0 && step + 1 < T) {
      if (bid != a.drop_bid) {
        const int so = slot_of(step);
        const int sp = slot_of(step + 2);
        // the poison stores of the previous step (and everything older) have left: see the protocol above
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane < 32) {                // lane (r, q' in {0,1}): units 8q'..8q'+7 of row r
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int p = 0; p < 3; ++p) {
              const f32x4 v = *reinterpret_cast<const f32x4*>(&L.hx[p][mt][r][(q & 1) * 8]);
              __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4v, v), xrs, xst + (mt * 3 + p) * 1024, so, 16);
            }
          if (!per_frame) {
            const u32x4v pz = {PERS_POISON, PERS_POISON, PERS_POISON, PERS_POISON};
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
              for (int p = 0; p < 3; ++p) __builtin_amdgcn_raw_buffer_store_b128(pz, xrs, xst + (mt * 3 + p) * 1024, sp, 16);
          }
        }
        PERS_STAMP(6);
      }
#pragma unroll
      for (int u = 0; u < NPRE; ++u) load(u, slot_of(step));
    }
    // the frame's outputs (activated gates, c, h)
#pragma unroll
    for (int i = 0; i < NEL; ++i) {
      if (el_ok[i]) {
        float* gp = a.gates + ((int64_t)t * N + el_n[i]) * H4 + j0 + r;
        gp[0] = go_[i][0];
        gp[H] = go_[i][1];
        gp[2 * H] = go_[i][2];
        gp[3 * H] = go_[i][3];
        a.c_all[((int64_t)t * N + el_n[i]) * H + j0 + r] = co_[i];
        reinterpret_cast<float*>(a.h_out)[((int64_t)t * N + el_n[i]) * a.ldh + j0 + r] = ho_[i];
      }
    }
  }
}

#endif      // DVAE_DEV (sentinel hand-off)

// ======================================================================================================================
// fp32 backward (the default arithmetic keeps the backward recurrence on the fp32 MFMA: it streams the 4H-wide gate
// gradients, which no operand split shrinks — lstm.hip).  Persistent form: a workgroup owns 16 hidden units
// x 32 segments; its 16 columns of W_hh over K = 4H are 64 K fp32 values per wave = 256 VGPRs per lane (fragments of
// v_mfma_f32_16x16x4_f32, dvae_lstm_pack_w), resident for the whole sequence.  dG[t] is handed over in fp32, one 1-KiB
// fragment (16 rows x 16 units) per (gate, row tile) and producer: a frame pulls 512 KB per CU from the exchange ring
// instead of 768 KB per workgroup (W_hh slice + rows) from L2 in the per-frame kernel.  Group = the H/16 workgroups of a row block.
// ======================================================================================================================
template <int KL>
struct F32BwdLds {
  f32x4 wl[KL > 0 ? NWV : 1][4][KL > 0 ? KL : 1][64];   // W_hh fragments kept in LDS: [wave][g][k - KR][lane]
  f32x4 red[NWV][2][64];                 // partial dH tiles [wave][mt][lane]
  float gx[4 * 2][16][20];               // dG[t] in fragment order [(g, mt)][row][16 units + pad]
  float bsum[4][16];
  int dead;
};

template <int H, int KL>
__global__ __launch_bounds__(64 * NWV, 1) void lstm_pers_bwd_f32(const PersArgs a) {
  constexpr int MT = 2, NEL = 2;
  constexpr int NPR = H / 16;           // producers of a row group = 16-deep k-chunks per gate
  constexpr int KW = NPR / NWV;         // chunks per gate and wave (its unit quarter)
  constexpr int KR = KW - KL;           // ... of which in registers (the fp32 fragments of H = 1024 are 256 registers per lane:
                                        // half of them live in LDS, 128 KB per workgroup)
  constexpr int NC = 4 * KW;            // chunks a wave contracts per frame
  constexpr int RD = PERS_RD_F32;       // chunks in flight (x 2 row tiles of 1 KiB each per lane-instruction)
  const int T = a.T, N = a.N;
  const int bid = blockIdx.x;
  const int rb = bid % a.n_rb, jb = bid / a.n_rb;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, q = lane >> 4;

  extern __shared__ __attribute__((aligned(16))) char lds_raw[];
  F32BwdLds<KL>& L = *reinterpret_cast<F32BwdLds<KL>*>(lds_raw);
  volatile int* dead = &L.dead;
  if (tid == 0) *dead = 0;
  const unsigned epoch = pers_epoch(a);
  if (tid < 64) L.bsum[tid >> 4][tid & 15] = 0.f;
  float bs[4] = {0.f, 0.f, 0.f, 0.f};

  // resident W_hh fragments: packed_bwd [(jb*4 + g)][chunk][lane][4] <- W[g*H + 16*chunk + 4q + e][jb*16 + r]
  f32x4 W[4][KR];
#pragma unroll
  for (int g = 0; g < 4; ++g)
#pragma unroll
    for (int k = 0; k < KW; ++k) {
      const f32x4 w = *reinterpret_cast<const f32x4*>(a.wp + ((((int64_t)(jb * 4 + g)) * NPR + wave * KW + k) * 64 + lane) * 16);
      if (k < KR) W[g][k < KR ? k : 0] = w;
      else L.wl[wave][g][k >= KR ? k - KR : 0][lane] = w;
    }

  // this wave owns (row tile emt, accumulator registers e0, e0+1) of every lane: rows emt*16 + q*4 + e0 + {0,1}, unit r
  const int emt = wave >> 1, e0 = (wave & 1) * 2;
  const int erow0 = emt * 16 + q * 4 + e0;
  int el_n[NEL];
  bool el_ok[NEL];
  float dcreg[NEL], ccreg[NEL];
#pragma unroll
  for (int i = 0; i < NEL; ++i) {
    el_ok[i] = rb * 32 + erow0 + i < N;
    el_n[i] = min(rb * 32 + erow0 + i, N - 1);
    dcreg[i] = 0.f;
  }
  const int j0 = jb * 16;
  const int64_t H4 = 4 * (int64_t)H;

  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)a.xch, 0, a.xch_bytes, 0x00020000);
  const int slot_bytes = a.n_rb * 4 * NPR * MT * 1024;                    // fragment (g, producer, mt) = 1 KiB
  const int xld = (rb * 4 * NPR + wave * KW) * MT * 1024 + lane * 16;     // + g*NPR*MT*1024 per gate
  const int xst = (rb * 4 * NPR + jb) * MT * 1024 + lane * 16;
  const unsigned* pflag = a.flags + rb * PERS_FLAG_LD_X3 + (lane < NPR ? lane : 0) * PERS_FLAG_STRIDE;
  unsigned* myflag = a.flags + rb * PERS_FLAG_LD_X3 + jb * PERS_FLAG_STRIDE;

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
      for (int g = 0; g < 4; ++g) o.gt[i][g] = a.gates[((int64_t)t_ * N + el_n[i]) * H4 + g * H + j0 + r];
      o.cp[i] = a.c_all[((int64_t)tp_ * N + el_n[i]) * H + j0 + r];
      o.dho[i] = a.dh_out[((int64_t)t_ * N + el_n[i]) * a.ldh + j0 + r];
    }
  };

  auto frame = [&](int step, Ops& cur, Ops& nxt) -> bool {
    const int fstep = T - 1 - step;
    const int t = frame_t(step);
    f32x4 acc[MT] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};

    PERS_STAMP(0);
    if (step > 0) {
      if (wave == NWV - 1 && !poll_ge(pflag, lane < NPR, epoch + (unsigned)step, a.timeout)) {
        pers_give_up(a.err, 2, bid, step, wave);
        *dead = 1;
      }
      __syncthreads();                                             // barrier A
      PERS_STAMP(1);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      const int so = ((step - 1) & 1) * slot_bytes;
      // chunk c of this wave: gate g = c / KW, producer k = c % KW of its unit quarter; RD chunks in flight
      f32x4 av[RD][MT];
      auto load = [&](int c) {
        const int g = c / KW, k = c % KW;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
          av[c % RD][mt] = __builtin_bit_cast(
              f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, xld + ((g * NPR + k) * MT + mt) * 1024, so, 16));
      };
#pragma unroll
      for (int c = 0; c < RD; ++c) load(c);
      fetch(min(step + 1, T - 1), nxt);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        const int g = c / KW, k = c % KW;
        const f32x4 w = (k < KR) ? W[g][k < KR ? k : 0] : L.wl[wave][g][k >= KR ? k - KR : 0][lane];
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
            acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[c % RD][mt][e], w[e], acc[mt], 0, 0, 0);
        if (c + RD < NC) {
          __builtin_amdgcn_sched_barrier(0);
          load(c + RD);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    } else {
      fetch(min(step + 1, T - 1), nxt);
    }
    L.red[wave][0][lane] = acc[0];
    L.red[wave][1][lane] = acc[1];
    PERS_STAMP(2);
    __syncthreads();                                               // barrier B
    PERS_STAMP(3);
    if (*dead) return false;

    float o_[NEL][4];
    {
      f32x2 rec = *reinterpret_cast<const f32x2*>(reinterpret_cast<const float*>(&L.red[0][emt][lane]) + e0);
#pragma unroll
      for (int w = 1; w < NWV; ++w)
        rec += *reinterpret_cast<const f32x2*>(reinterpret_cast<const float*>(&L.red[w][emt][lane]) + e0);
#pragma unroll
      for (int i = 0; i < NEL; ++i) {
        const float dh = cur.dho[i] + rec[i];
        const float gi = cur.gt[i][0], gf = cur.gt[i][1], gg = cur.gt[i][2], go = cur.gt[i][3];
        const float cp = fstep > 0 ? cur.cp[i] : 0.f;
        const float tc = gate_tanh(ccreg[i]);
        const float dc = dcreg[i] + dh * go * (1.f - tc * tc);
        o_[i][0] = dc * gg * gi * (1.f - gi);
        o_[i][1] = dc * cp * gf * (1.f - gf);
        o_[i][2] = dc * gi * (1.f - gg * gg);
        o_[i][3] = dh * tc * go * (1.f - go);
        dcreg[i] = dc * gf;
        ccreg[i] = cp;
        const int row = erow0 + i;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          L.gx[g * MT + (row >> 4)][row & 15][r] = o_[i][g];
          if (el_ok[i]) bs[g] += o_[i][g];
        }
      }
    }
    PERS_STAMP(4);
    __syncthreads();                                               // barrier C
    PERS_STAMP(5);
    if (wave == 0) {
      if ((step + 1 < T) && (bid != a.drop_bid)) {
        const int so = (step & 1) * slot_bytes;
        // lane (r, q): units 4q..4q+3 of row r = the 16 bytes of the fragment
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(&L.gx[g * MT + mt][r][q * 4]);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4v, v), xrs, xst + (g * NPR * MT + mt) * 1024, so, 16);
          }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        PERS_STAMP(6);
        if (lane == 0) __hip_atomic_store(myflag, epoch + (unsigned)(step + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    } else {
      // waves 1..3 archive dG[t] (fp32, 64-byte row pieces) for the weight-gradient / dx contractions
      for (int f = wave - 1; f < 4 * MT; f += NWV - 1) {
        const int g = f / MT, mt = f - g * MT;
        const int n = rb * 32 + mt * 16 + r;
        if (n < N)
          *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(a.dgates) + ((int64_t)t * N + n) * H4 + g * H + j0 + q * 4) =
              *reinterpret_cast<const f32x4*>(&L.gx[f][r][q * 4]);
      }
    }
    return true;
  };

  {
    const int t0 = frame_t(0);
#pragma unroll
    for (int i = 0; i < NEL; ++i) ccreg[i] = a.c_all[((int64_t)t0 * N + el_n[i]) * H + j0 + r];
  }
  Ops oa, ob;
  fetch(0, oa);
  __syncthreads();
  for (int step = 0; step < T; step += 2) {
    if (!frame(step, oa, ob)) break;
    if (step + 1 < T && !frame(step + 1, ob, oa)) break;
  }
  pers_bias_out<16>(a, lds_raw, bs, r, rb, j0, H);
  pers_finish(a);
}

// ======================================================================================================================
// fp32x3 backward (experiment → see DESIGN.md §4.2b): as lstm_pers_bwd_f32, but the recurrent product runs on the bf16 pipe
// — W_hh as three resident bf16 planes (planes 0, 1 in registers, plane 2 in LDS), dG[t+1] handed over in fp32 and split
// into three planes BY THE CONSUMER (each value is split by the 64 workgroups that read it: the split is VALU work in the
// shadow of the MFMAs, 44 operations per 8 values against six 16-cycle MFMAs).  fp32 results, 6/16 of the fp32-MFMA cycles.
// ======================================================================================================================
template <int K2L, int MT = 2>
struct X3BwdLds {
  bf16x8 w2[K2L > 0 ? NWV : 1][4][K2L > 0 ? K2L : 1][64];   // plane 2: [wave][g][k - K2R][lane]
  f32x4 red[NWV][MT][64];
  float gx[4 * MT][16][20];              // dG[t] (fp32) [(g, mt)][row][16 units + pad]
  float bsum[4][16];
  int dead;
  int loc;                               // pers_loc_check's verdict (frame 1)
};

template <int H, int K2L, int MT = 2>      // MT: see lstm_pers_fwd_x3
__global__ __launch_bounds__(64 * NWV, 1) void lstm_pers_bwd_x3(const PersArgs a) {
  constexpr int NEL = MT;
  constexpr int NCH = H / 32;           // 32-deep k-chunks per gate
  constexpr int NPR = H / 16;           // producers of a row group
  constexpr int KW = NCH / NWV;         // chunks per gate and wave
  constexpr int K2R = KW - K2L;
  constexpr int NU = 4 * KW * MT;       // (gate, chunk, row tile) units a wave contracts per frame
  constexpr int RD = NU < 8 ? NU : 8;   // units in flight (two 1-KiB loads per lane each)
  const int T = a.T, N = a.N;
  const int bid = blockIdx.x;
  const int rb = bid % a.n_rb, jb = bid / a.n_rb;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, q = lane >> 4;

  extern __shared__ __attribute__((aligned(16))) char lds_raw[];
  X3BwdLds<K2L, MT>& L = *reinterpret_cast<X3BwdLds<K2L, MT>*>(lds_raw);
  volatile int* dead = &L.dead;
  if (tid == 0) { *dead = 0; L.loc = 0; }
  const unsigned epoch = pers_epoch(a);
  const unsigned long long loc_tag = pers_loc_tag(epoch, pers_xcc());
  bool loc = false;                     // XCD-local hand-off from frame 1 on (pers_loc_*)
  if (tid < 64) L.bsum[tid >> 4][tid & 15] = 0.f;
  float bs[4] = {0.f, 0.f, 0.f, 0.f};

  // packed_bwd (three planes): [(jb*4 + g)][chunk][plane][lane][8] <- W[g*H + 32*chunk + 8q + j][jb*16 + r]
  bf16x8 W01[4][KW][2], W2[4][K2R > 0 ? K2R : 1];
#pragma unroll
  for (int g = 0; g < 4; ++g)
#pragma unroll
    for (int k = 0; k < KW; ++k)
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        const bf16x8 w = *reinterpret_cast<const bf16x8*>(
            a.wp + (((((int64_t)(jb * 4 + g)) * NCH + wave * KW + k) * 3 + p) * 64 + lane) * 16);
        if (p < 2) W01[g][k][p < 2 ? p : 0] = w;
        else if (k < K2R) W2[g][k < K2R ? k : 0] = w;
        else L.w2[wave][g][k >= K2R ? k - K2R : 0][lane] = w;
      }

  const int emt = MT == 2 ? (wave >> 1) : 0, e0 = MT == 2 ? (wave & 1) * 2 : wave;
  const int erow0 = emt * 16 + q * 4 + e0;
  int el_n[NEL];
  bool el_ok[NEL];
  float dcreg[NEL], ccreg[NEL];
#pragma unroll
  for (int i = 0; i < NEL; ++i) {
    el_ok[i] = rb * 16 * MT + erow0 + i < N;
    el_n[i] = min(rb * 16 * MT + erow0 + i, N - 1);
    dcreg[i] = 0.f;
  }
  const int j0 = jb * 16;
  const int64_t H4 = 4 * (int64_t)H;

  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)a.xch, 0, a.xch_bytes, 0x00020000);
  const int slot_bytes = a.n_rb * 4 * NCH * MT * 2048;                    // fragment (g, chunk, mt) = two 1-KiB halves
  const int xld = (rb * 4 * NCH + wave * KW) * MT * 2048 + lane * 16;     // + ((g*NCH + k)*MT + mt)*2048 + half*1024
  // this workgroup's 16 units are half of chunk jb/2: lanes 32*(jb&1) + (r + 16 q'), q' in {0,1}
  const int xst = (rb * 4 * NCH + (jb >> 1)) * MT * 2048 + ((jb & 1) * 32 + lane) * 16;
  const unsigned* pflag = a.flags + rb * PERS_FLAG_LD_X3 + (lane < NPR ? lane : 0) * PERS_FLAG_STRIDE;
  unsigned* myflag = a.flags + rb * PERS_FLAG_LD_X3 + jb * PERS_FLAG_STRIDE;

  struct Ops {
    float gt[NEL][4], cp[NEL], dho[NEL];
  };
  auto frame_t = [&](int step_) { const int fs = T - 1 - step_; return a.reverse ? (T - 1 - fs) : fs; };
  auto fetch = [&](int step_, Ops& o) __attribute__((always_inline)) {
    const int t_ = frame_t(step_);
    const int tp_ = min(max(a.reverse ? t_ + 1 : t_ - 1, 0), T - 1);
#pragma unroll
    for (int i = 0; i < NEL; ++i) {
#pragma unroll
      for (int g = 0; g < 4; ++g) o.gt[i][g] = a.gates[((int64_t)t_ * N + el_n[i]) * H4 + g * H + j0 + r];
      o.cp[i] = a.c_all[((int64_t)tp_ * N + el_n[i]) * H + j0 + r];
      o.dho[i] = a.dh_out[((int64_t)t_ * N + el_n[i]) * a.ldh + j0 + r];
    }
  };

  auto frame = [&](int step, Ops& cur, Ops& nxt) __attribute__((always_inline)) -> bool {
    const int fstep = T - 1 - step;
    const int t = frame_t(step);
    f32x4 acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};

    PERS_STAMP(0);
    if (step > 0) {
      if (wave == NWV - 1) {
        if (!poll_ge(pflag, lane < NPR, epoch + (unsigned)step, a.timeout)) {
          pers_give_up(a.err, 2, bid, step, wave);
          *dead = 1;
        } else if (a.local_ok && step == 1) {
          const bool ok = pers_loc_check(pflag, lane < NPR, loc_tag);
          if (lane == 0) L.loc = ok;
        }
      }
      __syncthreads();                                             // barrier A
      if (a.local_ok && step == 1) {
        loc = L.loc != 0;
        if (loc && bid == 0 && tid == 0) atomicAdd(a.err + PERS_LOCAL_WORD, 1u);
      }
      PERS_STAMP(1);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      const int so = ((step - 1) & 1) * slot_bytes;
      // unit u: gate g = u / (KW*MT), chunk k = (u / MT) % KW, row tile mt = u % MT; RD units in flight
      f32x4 av[RD][2];
      auto load = [&](int u) __attribute__((always_inline)) {
        const int g = u / (KW * MT), k = (u / MT) % KW, mt = u % MT;
        // the per-lane part of the address is ONE register (xld); everything else rides in the scalar offset
        const int off = so + ((g * NCH + k) * MT + mt) * 2048;
        av[u % RD][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, xld, off, 16));
        av[u % RD][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, xld, off + 1024, 16));
      };
      auto split = [&](int u, bf16x8 (&pl)[3]) __attribute__((always_inline)) {
        bf16x4 lo[3], hi[3];
        PERS_SPLIT3(av[u % RD][0], lo);
        PERS_SPLIT3(av[u % RD][1], hi);
#pragma unroll
        for (int p = 0; p < 3; ++p) pl[p] = __builtin_shufflevector(lo[p], hi[p], 0, 1, 2, 3, 4, 5, 6, 7);
      };
#pragma unroll
      for (int u = 0; u < RD; ++u) load(u);
      fetch(min(step + 1, T - 1), nxt);
      __builtin_amdgcn_sched_barrier(0);
      bf16x8 pl[2][3];                         // planes of the unit being multiplied / of the next one (split under its MFMAs)
      split(0, pl[0]);
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int k = 0; k < KW; ++k) {
          const bf16x8 w2 = (k < K2R) ? W2[g][k < K2R ? k : 0] : L.w2[wave][g][k >= K2R ? k - K2R : 0][lane];
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
            const int u = (g * KW + k) * MT + mt;
            if (u + 1 < NU) split(u + 1, pl[(u + 1) & 1]);
            const bf16x8 (&pc)[3] = pl[u & 1];
            acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pc[0], W01[g][k][0], acc[mt], 0, 0, 0);
            acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pc[0], W01[g][k][1], acc[mt], 0, 0, 0);
            acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pc[1], W01[g][k][0], acc[mt], 0, 0, 0);
            acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pc[1], W01[g][k][1], acc[mt], 0, 0, 0);
            acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pc[0], w2, acc[mt], 0, 0, 0);
            acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pc[2], W01[g][k][0], acc[mt], 0, 0, 0);
            if (u + RD < NU) {
              __builtin_amdgcn_sched_barrier(0);
              load(u + RD);
              __builtin_amdgcn_sched_barrier(0);
            }
          }
        }
    } else {
      fetch(min(step + 1, T - 1), nxt);
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) L.red[wave][mt][lane] = acc[mt];
    PERS_STAMP(2);
    __syncthreads();                                               // barrier B
    PERS_STAMP(3);
    if (*dead) return false;

    {
      float rec[NEL];
#pragma unroll
      for (int i = 0; i < NEL; ++i) rec[i] = reinterpret_cast<const float*>(&L.red[0][emt][lane])[e0 + i];
#pragma unroll
      for (int w = 1; w < NWV; ++w)
#pragma unroll
        for (int i = 0; i < NEL; ++i) rec[i] += reinterpret_cast<const float*>(&L.red[w][emt][lane])[e0 + i];
#pragma unroll
      for (int i = 0; i < NEL; ++i) {
        const float dh = cur.dho[i] + rec[i];
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
        ccreg[i] = cp;
        const int row = erow0 + i;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          L.gx[g * MT + (row >> 4)][row & 15][r] = o[g];
          if (el_ok[i]) bs[g] += o[g];
        }
      }
    }
    PERS_STAMP(4);
    __syncthreads();                                               // barrier C
    PERS_STAMP(5);
    if (wave == 0) {
      if ((step + 1 < T) && (bid != a.drop_bid)) {
        const int so = (step & 1) * slot_bytes;
        const bool lp = pers_loc_step(loc, step, T);
        if (a.local_ok && step == 0 && lane == 0) pers_loc_announce(myflag, loc_tag);
        if (lane < 32) {               // lane (r, q' in {0,1}): units 8q' + 4h .. + 3 of row r go to half h
#pragma unroll
          for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
              for (int h = 0; h < 2; ++h) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(&L.gx[g * MT + mt][r][(q & 1) * 8 + 4 * h]);
                PERS_ST(lp, __builtin_bit_cast(u32x4v, v), xrs, xst, so + ((g * NCH * MT) + mt) * 2048 + h * 1024);
              }
        }
        PERS_DRAIN();
        PERS_STAMP(6);
        if (lane == 0) PERS_FLAG(lp, myflag, epoch + (unsigned)(step + 1));
      }
    } else {
      for (int f = wave - 1; f < 4 * MT; f += NWV - 1) {
        const int g = f / MT, mt = f - g * MT;
        const int n = rb * 16 * MT + mt * 16 + r;
        if (n < N)
          *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(a.dgates) + ((int64_t)t * N + n) * H4 + g * H + j0 + q * 4) =
              *reinterpret_cast<const f32x4*>(&L.gx[f][r][q * 4]);
      }
    }
    return true;
  };

  {
    const int t0 = frame_t(0);
#pragma unroll
    for (int i = 0; i < NEL; ++i) ccreg[i] = a.c_all[((int64_t)t0 * N + el_n[i]) * H + j0 + r];
  }
  Ops oa, ob;
  fetch(0, oa);
  __syncthreads();
  for (int step = 0; step < T; step += 2) {
    if (!frame(step, oa, ob)) break;
    if (step + 1 < T && !frame(step + 1, ob, oa)) break;
  }
  if (loc && *dead && wave == 0) {      // gave up with lines in the L2: both slots' pieces and the flag once more, write-through
    const u32x4v z = {0u, 0u, 0u, 0u};
    if (lane < 32) {
#pragma unroll
      for (int sl = 0; sl < 2; ++sl)
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int h = 0; h < 2; ++h)
              __builtin_amdgcn_raw_buffer_store_b128(z, xrs, xst, sl * slot_bytes + ((g * NCH * MT) + mt) * 2048 + h * 1024, 16);
    }
    if (lane == 0) pers_loc_scrub_flag(myflag);
  }
  pers_bias_out<16>(a, lds_raw, bs, r, rb, j0, H);
  pers_finish(a);
}

// ======================================================================================================================
// fp32x3 backward, K-SPLIT form (round 4).  lstm_pers_bwd_x3 gives a workgroup 16 hidden units over ALL of K = 4H: every one
// of the 64 workgroups of a row group pulls the whole 32 x 4H fp32 slab of dG[t+1] (512 KB at H = 1024) through its L2 -> CU
// path and splits every value it loads (44 VALU operations per 8 values): 7.3 us of loads and 5.9 us of VALU issue per
// frame against 3.2 us of MFMA issue.  Here a workgroup owns 64 hidden units x ONE QUARTER of K (the gate columns of the
// hidden units [kq H/4, (kq+1) H/4), all four gates): the same W_hh bytes on chip and the same MFMAs, but a quarter of the dG
// rows (128 KB) and a quarter of the splits per CU — each loaded fragment now feeds four n-tiles (24 MFMAs instead of 6).
// The price is a second hand-off per frame: the four k-quarter workgroups of a (row group, 64-unit block) each hold a
// PARTIAL dh[32 x 64]; workgroup kq finishes the 16 units [16 kq, 16 kq + 16) of the block, so every workgroup sends three
// 32 x 16 tiles (2 KB each, one per peer, each with a flag of its own) and sums the three it receives with its own in the
// fixed order kq = 0..3.  From there the frame is lstm_pers_bwd_x3's: gate-derivative epilogue on 32 x 16 elements, dG[t] of
// those 16 units published in the same fragment order, same flags (producer index jb = 4 ub + kq = bid / n_rb).
//   wave w of workgroup (rb, ub, kq): k-chunks 8 kq + 2 w, + 1 of every gate (H = 1024), all four 16-unit n-tiles of ub.
//   exchange slot = [dG fragments as before | inbox: [workgroup][sender kq][row tile][lane] f32x4]
//   flags: dG flags as before; partial flags behind them, one 32-byte granule per (receiver, sender).
// ======================================================================================================================
constexpr int PERS_PFLAG_OFF = 4 * PERS_FLAG_LD_X3;      // words: partial flags start behind the dG flags of 4 row groups
constexpr int PERS_PFLAG_STRIDE = 8;                     // words between two partial flags
static_assert((PERS_PFLAG_OFF + 256 * 4 * PERS_PFLAG_STRIDE) * 4 <= PERS_FLAG_BYTES, "partial flags of 256 workgroups fit");

struct X3KLds {
  bf16x8 w2[NWV][4][4][2][64];   // plane 2 of W_hh: [wave][n-tile][gate][chunk][lane]  (H = 1024: 128 KB)
  union {
    f32x4 red[NWV][3][2][64];    // cross-wave reduction: wave w's partial tiles for the three n-tiles it does not finish
    struct {
      f32x4 fin[4][2][64];       // the four partial dh tiles of this workgroup's 16 units: [sender kq][row tile][lane]
      float gx[4 * 2][16][20];   // dG[t] (fp32) [(g, mt)][row][16 units + pad]
    } e;
  } u;
  float bsum[4][16];
  int dead;
};

template <int H>
__global__ __launch_bounds__(64 * NWV, 1) void lstm_pers_bwd_x3k(const PersArgs a) {
  constexpr int MT = 2, NEL = 2;
  constexpr int NCH = H / 32;           // 32-deep k-chunks per gate
  constexpr int KQ = NCH / 4;           // chunks per gate of one k-quarter
  constexpr int KW = KQ / NWV;          // chunks per gate and wave (2 at H = 1024, 1 at H = 512)
  constexpr int NU = 4 * KW * MT;       // (gate, chunk, row tile) units a wave loads per frame
#ifndef PERS_RD_X3K
#define PERS_RD_X3K 6
#endif
  constexpr int RD = NU < PERS_RD_X3K ? NU : PERS_RD_X3K;   // units in flight (two 1-KiB loads per lane each)
  static_assert(KW >= 1 && KW <= 2, "H = 512 or 1024");
  const int T = a.T, N = a.N;
  const int bid = blockIdx.x;
  const int rb = bid % a.n_rb, jb = bid / a.n_rb;      // jb = 4 ub + kq: the 16-unit block this workgroup FINISHES
  const int kq = jb & 3, ub = jb >> 2;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, q = lane >> 4;

  extern __shared__ __attribute__((aligned(16))) char lds_raw[];
  X3KLds& L = *reinterpret_cast<X3KLds*>(lds_raw);
  volatile int* dead = &L.dead;
  if (tid == 0) *dead = 0;
  const unsigned epoch = pers_epoch(a);
  if (tid < 64) L.bsum[tid >> 4][tid & 15] = 0.f;
  float bs[4] = {0.f, 0.f, 0.f, 0.f};

  // packed_bwd (three planes): [(jb'*4 + g)][chunk][plane][lane][8] <- W[g*H + 32*chunk + 8q + j][jb'*16 + r]
  bf16x8 W01[4][4][KW][2];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt)
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int k = 0; k < KW; ++k)
#pragma unroll
        for (int p = 0; p < 3; ++p) {
          const bf16x8 w = *reinterpret_cast<const bf16x8*>(
              a.wp + (((((int64_t)((ub * 4 + nt) * 4 + g)) * NCH + kq * KQ + wave * KW + k) * 3 + p) * 64 + lane) * 16);
          if (p < 2) W01[nt][g][k][p < 2 ? p : 0] = w;
          else L.w2[wave][nt][g][k][lane] = w;
        }

  const int emt = wave >> 1, e0 = (wave & 1) * 2;
  const int erow0 = emt * 16 + q * 4 + e0;
  int el_n[NEL];
  bool el_ok[NEL];
  float dcreg[NEL], ccreg[NEL];
#pragma unroll
  for (int i = 0; i < NEL; ++i) {
    el_ok[i] = rb * 32 + erow0 + i < N;
    el_n[i] = min(rb * 32 + erow0 + i, N - 1);
    dcreg[i] = 0.f;
  }
  const int j0 = jb * 16;
  const int64_t H4 = 4 * (int64_t)H;

  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)a.xch, 0, a.xch_bytes, 0x00020000);
  const int dg_bytes = a.n_rb * 4 * NCH * MT * 2048;                      // dG part of a slot
  const int slot_bytes = dg_bytes + (int)gridDim.x * 4 * MT * 1024;       // + inbox: [workgroup][sender][mt][lane] f32x4
  const int xld = (rb * 4 * NCH + kq * KQ + wave * KW) * MT * 2048 + lane * 16;     // + ((g*NCH + k)*MT + mt)*2048 + half*1024
  // this workgroup's 16 units are half of chunk jb/2: lanes 32*(jb&1) + (r + 16 q'), q' in {0,1}
  const int xst = (rb * 4 * NCH + (jb >> 1)) * MT * 2048 + ((jb & 1) * 32 + lane) * 16;
  // the dG of this k-quarter comes from the NPQ = H/64 workgroups that finish its units: jb' = NPQ kq .. NPQ kq + NPQ - 1
  constexpr int NPQ = H / 64;
  const unsigned* pflag = a.flags + rb * PERS_FLAG_LD_X3 + (NPQ * kq + (lane < NPQ ? lane : 0)) * PERS_FLAG_STRIDE;
  unsigned* myflag = a.flags + rb * PERS_FLAG_LD_X3 + jb * PERS_FLAG_STRIDE;
  // partial tiles: wave w != kq sends n-tile w to the peer that finishes it, bid_w = rb + n_rb (4 ub + w)
  const int peer_bid = rb + a.n_rb * (ub * 4 + wave);
  const int inb_st = dg_bytes + ((peer_bid * 4 + kq) * MT) * 1024 + lane * 16;     // + mt * 1024   (peer's inbox, sender = me)
  const int inb_ld = dg_bytes + ((bid * 4 + wave) * MT) * 1024 + lane * 16;        // + mt * 1024   (my inbox, sender = wave)
  unsigned* pf_out = a.flags + PERS_PFLAG_OFF + (peer_bid * 4 + kq) * PERS_PFLAG_STRIDE;
  const unsigned* pf_in = a.flags + PERS_PFLAG_OFF + (bid * 4 + (lane < 4 ? lane : 0)) * PERS_PFLAG_STRIDE;

  struct Ops {
    float gt[NEL][4], cp[NEL], dho[NEL];
  };
  auto frame_t = [&](int step_) { const int fs = T - 1 - step_; return a.reverse ? (T - 1 - fs) : fs; };
  auto fetch = [&](int step_, Ops& o) __attribute__((always_inline)) {
    const int t_ = frame_t(step_);
    const int tp_ = min(max(a.reverse ? t_ + 1 : t_ - 1, 0), T - 1);
#pragma unroll
    for (int i = 0; i < NEL; ++i) {
#pragma unroll
      for (int g = 0; g < 4; ++g) o.gt[i][g] = a.gates[((int64_t)t_ * N + el_n[i]) * H4 + g * H + j0 + r];
      o.cp[i] = a.c_all[((int64_t)tp_ * N + el_n[i]) * H + j0 + r];
      o.dho[i] = a.dh_out[((int64_t)t_ * N + el_n[i]) * a.ldh + j0 + r];
    }
  };

  auto frame = [&](int step, Ops& cur, Ops& nxt) __attribute__((always_inline)) -> bool {
    const int fstep = T - 1 - step;
    const int t = frame_t(step);
    PERS_STAMP(0);
    if (step > 0) {
      f32x4 acc[4][MT];
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[nt][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (wave == NWV - 1 && !poll_ge(pflag, lane < NPQ, epoch + (unsigned)step, a.timeout)) {
        pers_give_up(a.err, 2, bid, step, wave);
        *dead = 1;
      }
      __syncthreads();                                             // barrier A
      PERS_STAMP(1);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      const int so = ((step - 1) & 1) * slot_bytes;
      // unit u: gate g = u / (KW*MT), chunk k = (u / MT) % KW, row tile mt = u % MT; RD units in flight
      f32x4 av[RD][2];
      auto load = [&](int u) __attribute__((always_inline)) {
        const int g = u / (KW * MT), k = (u / MT) % KW, mt = u % MT;
        const int off = so + ((g * NCH + k) * MT + mt) * 2048;
        av[u % RD][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, xld, off, 16));
        av[u % RD][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, xld, off + 1024, 16));
      };
      auto split = [&](int u, bf16x8 (&pl)[3]) __attribute__((always_inline)) {
        bf16x4 lo[3], hi[3];
        PERS_SPLIT3(av[u % RD][0], lo);
        PERS_SPLIT3(av[u % RD][1], hi);
#pragma unroll
        for (int p = 0; p < 3; ++p) pl[p] = __builtin_shufflevector(lo[p], hi[p], 0, 1, 2, 3, 4, 5, 6, 7);
      };
#pragma unroll
      for (int u = 0; u < RD; ++u) load(u);
      __builtin_amdgcn_sched_barrier(0);
      bf16x8 pl[2][3];                         // planes of the unit being multiplied / of the next one (split under its MFMAs)
      split(0, pl[0]);
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int k = 0; k < KW; ++k)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
            const int u = (g * KW + k) * MT + mt;
            if (u + 1 < NU) split(u + 1, pl[(u + 1) & 1]);
            const bf16x8 (&pc)[3] = pl[u & 1];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
              const bf16x8 w2 = L.w2[wave][nt][g][k][lane];
              acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pc[0], W01[nt][g][k][0], acc[nt][mt], 0, 0, 0);
              acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pc[0], W01[nt][g][k][1], acc[nt][mt], 0, 0, 0);
              acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pc[1], W01[nt][g][k][0], acc[nt][mt], 0, 0, 0);
              acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pc[1], W01[nt][g][k][1], acc[nt][mt], 0, 0, 0);
              acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pc[0], w2, acc[nt][mt], 0, 0, 0);
              acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pc[2], W01[nt][g][k][0], acc[nt][mt], 0, 0, 0);
            }
            if (u + RD < NU) {
              __builtin_amdgcn_sched_barrier(0);
              load(u + RD);
              __builtin_amdgcn_sched_barrier(0);
            }
          }
      // next frame's epilogue operands: in flight across the two hand-offs below (fetched here, not under the MFMAs, where
      // their 12 registers made the loop spill)
      fetch(min(step + 1, T - 1), nxt);
      // ---- cross-wave reduction: wave w finishes n-tile w; its partial tiles of the other three go to LDS
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        if (nt == wave) continue;
        const int sidx = wave < nt ? wave : wave - 1;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) L.u.red[nt][sidx][mt][lane] = acc[nt][mt];
      }
      PERS_STAMP(2);
      __syncthreads();                                             // barrier B
      f32x4 mine[MT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        // fixed order: waves 0..3 (own partial in its place)
        f32x4 s4 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int w = 0; w < NWV; ++w) {
          f32x4 v;
          if (w == wave) {
            v = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) v = (nt == wave) ? acc[nt][mt] : v;
          } else {
            v = L.u.red[wave][w < wave ? w : w - 1][mt][lane];
          }
          s4 += v;
        }
        mine[mt] = s4;
      }
      // ---- partial hand-off: n-tile `wave` belongs to peer kq' = wave; the workgroup's own tile stays
      if (wave != kq) {
        if (bid != a.drop_bid) {
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4v, mine[mt]), xrs, inb_st, so + mt * 1024, 16);
          PERS_DRAIN();
          if (lane == 0) __hip_atomic_store(pf_out, epoch + (unsigned)step, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      PERS_STAMP(3);
      __syncthreads();                 // barrier C: `red` is dead from here (fin / gx alias it)
      if (wave == kq) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) L.u.e.fin[kq][mt][lane] = mine[mt];
      }
      if (wave == NWV - 1 && !poll_ge(pf_in, lane < 4 && lane != kq, epoch + (unsigned)step, a.timeout)) {
        pers_give_up(a.err, 2, bid, step, wave);
        *dead = 1;
      }
      __syncthreads();                                             // barrier D
      PERS_STAMP(4);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      if (wave != kq && !*dead) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
          L.u.e.fin[wave][mt][lane] =
              __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, inb_ld, so + mt * 1024, 16));
      }
    } else {
      fetch(min(step + 1, T - 1), nxt);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) L.u.e.fin[wave][mt][lane] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    __syncthreads();                                               // barrier E
    PERS_STAMP(5);
    if (*dead) return false;

    {
      f32x2 rec = *reinterpret_cast<const f32x2*>(reinterpret_cast<const float*>(&L.u.e.fin[0][emt][lane]) + e0);
#pragma unroll
      for (int w = 1; w < 4; ++w)
        rec += *reinterpret_cast<const f32x2*>(reinterpret_cast<const float*>(&L.u.e.fin[w][emt][lane]) + e0);
#pragma unroll
      for (int i = 0; i < NEL; ++i) {
        const float dh = cur.dho[i] + rec[i];
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
        ccreg[i] = cp;
        const int row = erow0 + i;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          L.u.e.gx[g * MT + (row >> 4)][row & 15][r] = o[g];
          if (el_ok[i]) bs[g] += o[g];
        }
      }
    }
    __syncthreads();                                               // barrier F
    PERS_STAMP(6);
    if (wave == 0) {
      if ((step + 1 < T) && (bid != a.drop_bid)) {
        const int so = (step & 1) * slot_bytes;
        if (lane < 32) {               // lane (r, q' in {0,1}): units 8q' + 4h .. + 3 of row r go to half h
#pragma unroll
          for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
              for (int h = 0; h < 2; ++h) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(&L.u.e.gx[g * MT + mt][r][(q & 1) * 8 + 4 * h]);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4v, v), xrs, xst,
                                                       so + ((g * NCH * MT) + mt) * 2048 + h * 1024, 16);
              }
        }
        PERS_DRAIN();
        PERS_STAMP(7);
        if (lane == 0) __hip_atomic_store(myflag, epoch + (unsigned)(step + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    } else {
      for (int f = wave - 1; f < 4 * MT; f += NWV - 1) {
        const int g = f / MT, mt = f - g * MT;
        const int n = rb * 32 + mt * 16 + r;
        if (n < N)
          *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(a.dgates) + ((int64_t)t * N + n) * H4 + g * H + j0 + q * 4) =
              *reinterpret_cast<const f32x4*>(&L.u.e.gx[f][r][q * 4]);
      }
    }
    // (gx / fin are rewritten only behind the next frame's barriers A..C; `red` aliases them and is written after barrier A)
    return true;
  };

  {
    const int t0 = frame_t(0);
#pragma unroll
    for (int i = 0; i < NEL; ++i) ccreg[i] = a.c_all[((int64_t)t0 * N + el_n[i]) * H + j0 + r];
  }
  Ops oa, ob;
  fetch(0, oa);
  __syncthreads();
  for (int step = 0; step < T; step += 2) {
    if (!frame(step, oa, ob)) break;
    if (step + 1 < T && !frame(step + 1, ob, oa)) break;
  }
  pers_bias_out<16>(a, lds_raw, bs, r, rb, j0, H);
  pers_finish(a);
}

#ifdef DVAE_PERS_TS
unsigned long long* g_pers_ts = nullptr;
int g_pers_ts_bid = 0;
#endif
#ifdef DVAE_DEV
unsigned* g_pers_dbg = nullptr;
unsigned* g_pers_dbg_frag = nullptr;
int g_pers_dbg_jb = 0, g_pers_nslot = 2;
#endif
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

#ifndef PERS_KL
#define PERS_KL 2
#endif

#ifdef DVAE_DEV
// sentinel kernels: the first two ring slots are poisoned in front of the launch (write-through 16-byte stores; every other
// slot by the producers themselves)
__global__ __launch_bounds__(256) void pers_poison_kernel(char* __restrict__ xch, int bytes) {
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)xch, 0, bytes, 0x00020000);
  const u32x4v pz = {PERS_POISON, PERS_POISON, PERS_POISON, PERS_POISON};
  for (int o = (blockIdx.x * 256 + threadIdx.x) * 16; o < bytes; o += gridDim.x * 256 * 16)
    __builtin_amdgcn_raw_buffer_store_b128(pz, rs, o, 0, 16);
}

#endif

// one launch: LDS padded so that exactly one workgroup fits a CU
template <class K>
int pers_go(K kern, int need_lds, const PersArgs& a, int grid, hipStream_t s) {
  const int lds = need_lds > PERS_PAD_LDS ? need_lds : PERS_PAD_LDS;
  (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);   // idempotent, host-side only
  hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * NWV), lds, s, a);
  return dvae_check_launch();
}

// kind: 0 bf16 forward, 1 bf16 backward, 2 fp32x3 forward, 3 fp32 backward, 4 fp32x3 backward, 5 fp32x3 backward (k-split),
// 6 fp32x3 forward with 8 units per workgroup
int pers_dispatch(int kind, int H, int mt, const PersArgs& a, int grid, hipStream_t s) {
  constexpr int KL = PERS_KL;
  switch (kind) {
    case 0:
      if (H == 1024 && mt == 1) return pers_go(lstm_pers_fwd_bf16<1024, 1, 0>, (int)sizeof(FwdLds<1, 0>), a, grid, s);
      if (H == 1024) return pers_go(lstm_pers_fwd_bf16<1024, 2, KL>, (int)sizeof(FwdLds<2, KL>), a, grid, s);
      if (mt == 1) return pers_go(lstm_pers_fwd_bf16<512, 1, 0>, (int)sizeof(FwdLds<1, 0>), a, grid, s);
      return pers_go(lstm_pers_fwd_bf16<512, 2, 0>, (int)sizeof(FwdLds<2, 0>), a, grid, s);
    case 1:
      if (H == 1024 && mt == 1) return pers_go(lstm_pers_bwd_bf16<1024, 1, 0>, (int)sizeof(BwdLds<1, 0>), a, grid, s);
      if (H == 1024) return pers_go(lstm_pers_bwd_bf16<1024, 2, KL>, (int)sizeof(BwdLds<2, KL>), a, grid, s);
      if (mt == 1) return pers_go(lstm_pers_bwd_bf16<512, 1, 0>, (int)sizeof(BwdLds<1, 0>), a, grid, s);
      return pers_go(lstm_pers_bwd_bf16<512, 2, 0>, (int)sizeof(BwdLds<2, 0>), a, grid, s);
#ifdef DVAE_DEV
    case 12:     // fp32x3 forward, sentinel hand-off
      if (H == 1024) return pers_go(lstm_pers_fwd_x3s<1024, 8>, (int)sizeof(X3Lds<8, 8>), a, grid, s);
      if (mt == 1) return pers_go(lstm_pers_fwd_x3s<512, 0, 1>, (int)sizeof(X3Lds<4, 0, 1>), a, grid, s);
      return pers_go(lstm_pers_fwd_x3s<512, 0>, (int)sizeof(X3Lds<4, 0>), a, grid, s);
#endif
    case 2:
      if (H == 1024) return pers_go(lstm_pers_fwd_x3<1024, 8>, (int)sizeof(X3Lds<8, 8>), a, grid, s);
#ifdef DVAE_DEV      // (the 16-row forward form is not in the product library: see dvae_pers_launch)
      if (mt == 1) return pers_go(lstm_pers_fwd_x3<512, 0, 1>, (int)sizeof(X3Lds<4, 0, 1>), a, grid, s);
#endif
      if (mt == 1) return DVAE_EINVAL;
      return pers_go(lstm_pers_fwd_x3<512, 0>, (int)sizeof(X3Lds<4, 0>), a, grid, s);
    case 6:      // fp32x3 forward, 8 units x 32 rows (H = 512)
      return pers_go(lstm_pers_fwd_x3h<512>, (int)sizeof(X3HLds), a, grid, s);
    case 3:
      if (H == 1024) return pers_go(lstm_pers_bwd_f32<1024, 8>, (int)sizeof(F32BwdLds<8>), a, grid, s);
      return pers_go(lstm_pers_bwd_f32<512, 0>, (int)sizeof(F32BwdLds<0>), a, grid, s);
    case 5:
      if (H == 1024) return pers_go(lstm_pers_bwd_x3k<1024>, (int)sizeof(X3KLds), a, grid, s);
      return pers_go(lstm_pers_bwd_x3k<512>, (int)sizeof(X3KLds), a, grid, s);
    default:
      if (H == 1024) return pers_go(lstm_pers_bwd_x3<1024, 8>, (int)sizeof(X3BwdLds<8>), a, grid, s);
      if (mt == 1) return pers_go(lstm_pers_bwd_x3<512, 0, 1>, (int)sizeof(X3BwdLds<0, 1>), a, grid, s);
      return pers_go(lstm_pers_bwd_x3<512, 0>, (int)sizeof(X3BwdLds<0>), a, grid, s);
  }
}

// fp32x3 / fp32 kernels: 32-row tiles, 16 units per workgroup
bool pers_x3_ok(int N, int H, int cus) {
  if (H != 512 && H != 1024) return false;
  const int n_rb = (N + 31) / 32;
  return n_rb <= 4 && (H / 16) * n_rb <= cus;
}

// bytes of ONE exchange slot (the fragments all workgroups publish in one frame) of kernel `kind`
int64_t pers_slot_bytes(int kind, int N, int H, int mt) {
  switch (kind) {
    case 0: return (int64_t)((N + 16 * mt - 1) / (16 * mt)) * (H / 32) * mt * 1024;
    case 1: return (int64_t)((N + 16 * mt - 1) / (16 * mt)) * 4 * (H / 32) * mt * 1024;
    case 2: return (int64_t)((N + 16 * mt - 1) / (16 * mt)) * (H / 32) * mt * 3 * 1024;
    case 3: return (int64_t)((N + 31) / 32) * 4 * (H / 16) * 2 * 1024;
    case 5: return (int64_t)((N + 31) / 32) * 4 * (H / 32) * 2 * 2048 +            // dG fragments
                   (int64_t)((N + 31) / 32) * (H / 16) * 4 * 2 * 1024;             // + every workgroup's inbox of partial tiles
    default: return (int64_t)((N + 16 * mt - 1) / (16 * mt)) * 4 * (H / 32) * mt * 2048;
  }
}

}  // namespace

// fp32x3 backward: 1 when the k-split kernel (lstm_pers_bwd_x3k) runs for this H — H = 1024 (measured 12.7 -> 9.85 us per
// frame; H = 512: 6.74 -> 6.56, not worth a second hand-off).  DVAE_PERS_BWD_KSPLIT (dev build): bit 0 H = 1024, bit 1 H = 512.
int dvae_pers_bwd_ksplit(int H) {
  static const int ksplit = dvae_dev_knob("DVAE_PERS_BWD_KSPLIT", 1);
  return ((H == 1024 && (ksplit & 1)) || (H == 512 && (ksplit & 2))) ? 1 : 0;
}

// fp32x3 forward: hidden units per workgroup — 8 (lstm_pers_fwd_x3h) at H = 512 where 64 workgroups per row group fit the chip,
// else 16 (lstm_pers_fwd_x3).  Also read by lstm.hip's byte accounting.
int dvae_pers_fwd_units(int N, int H) {
  static const int h8 = dvae_dev_knob("DVAE_PERS_X3_H8", 1);
  return (h8 && H == 512 && (H / 8) * ((N + 31) / 32) <= pers_cu_count()) ? 8 : 16;
}

// used by lstm.hip: 1 when (N, H, mode, pass) has a persistent kernel on this device
int dvae_pers_usable(int N, int H, int pm, int bwd) {
  if (pm == DVAE_MODE_F32X3) return pers_x3_ok(N, H, pers_cu_count());      // forward, and backward with consumer-side split
  if (pm == DVAE_MODE_F32) return bwd && pers_x3_ok(N, H, pers_cu_count());      // fp32 backward: the same 16 x 32 tiling
  if (pm != DVAE_MODE_BF16) return 0;
  return pers_mt(N, H, pers_cu_count()) != 0;
}

DVAE_API int dvae_lstm_pers_supported(int N, int H, int mode, int bwd) {
  if (N < 1) return 0;
  return dvae_pers_usable(N, H, mode, bwd);
}

static int64_t pers_ws_need(int T, int N, int H) {
  if (N < 1 || (H != 512 && H != 1024)) return 0;
  int64_t slot = 0;
  if (int mt = pers_mt(N, H, 256))
    for (int kind = 0; kind < 2; ++kind) slot = std::max(slot, pers_slot_bytes(kind, N, H, mt));
  if (pers_x3_ok(N, H, 256))
    for (int kind = 2; kind < 6; ++kind) slot = std::max(slot, pers_slot_bytes(kind, N, H, 2));
  if (!slot) return 0;
  return PERS_XCH_OFF + (T > 0 ? (int64_t)T : 4) * slot;      // (two slots in use; four sized: the dev build's sentinel form)
}
DVAE_API int64_t dvae_lstm_pers_ws_bytes(int N, int H) { return pers_ws_need(0, N, H); }

// one direction of one layer, all T frames; `bwd` selects the pass.  Returns DVAE_EINVAL when the shape has no persistent
// kernel (the caller then uses the per-frame launches).
int dvae_pers_launch(const dvae_lstm_dir_t& d, bool bwd, int T, int N, int H, int64_t ldh, int drop_bid, hipStream_t s) {
  const int cus = pers_cu_count();
  if (!d.pers_ws || !d.w_packed || (((uintptr_t)d.pers_ws) & 255)) return DVAE_EINVAL;
  int kind, mt = 2;
  bool units8 = false;
  if (d.packed_mode == DVAE_MODE_BF16) {
    mt = pers_mt(N, H, cus);
    if (!mt) return DVAE_EINVAL;
    if (d.state_bf16 && (ldh & 7)) return DVAE_EINVAL;
    kind = bwd ? 1 : 0;
  } else {
    if (d.state_bf16 || !pers_x3_ok(N, H, cus) || (ldh & 3)) return DVAE_EINVAL;
    // backward: the k-split form at H = 1024 (a quarter of the dG rows per CU); DVAE_PERS_BWD_KSPLIT
    // (dev build) picks per H: bit 0 H = 1024, bit 1 H = 512
    const bool ks = bwd && dvae_pers_bwd_ksplit(H);
    if (d.packed_mode == DVAE_MODE_F32X3) kind = bwd ? (ks ? 5 : 4) : 2;
    else if (d.packed_mode == DVAE_MODE_F32 && bwd) kind = 3;
    else return DVAE_EINVAL;
    // 16-row tiles where 32-row tiles would leave CUs idle (H = 512, N <= 128: 32 x 8 workgroups instead of 32 x 4);
    // DVAE_PERS_X3_MT1=0 (dev build) keeps 32 rows
    // bit 1: backward (measured 6.75 -> 4.66 us per frame; 0 failures in 300 rounds of scripts/x3_handoff_stress.py), bit 0:
    // forward — DEV BUILD ONLY: 4.43 -> 3.41 us per frame, but 5-17 % of the stress rounds (a second stream streaming GiBs
    // through HBM meanwhile) came back with rows 12..15 of one row group wrong in ALL columns from a frame around the 80th
    // microsecond on.  Not the hand-off protocol (a release fence in front of the flag, an agent-scope acquire in front of
    // the loads, sc0 sc1 loads, four ring slots, row groups spread over XCDs, a full vmcnt(0) after the loads: each still
    // failed) and not found by reading the ISA; the 32-row forward kernel and every other persistent kernel: 0 of 200.
    // scripts/x3_fwd16_diag.py: the stored h[t-1] of the bad rows is right and the bad gates match NO candidate input (h of
    // another frame, another row, zeros) — in 3 of 4 cases they are off by more than any |h| < 1 could move them: the
    // consumers' fragments held foreign bits in the 64-byte pieces of rows 12..15 (a fragment row group = lanes 12..15 of
    // each 16-lane quarter), e.g. words an earlier launch left in the ring; once they were merely slightly off (1e-2).
    static const int mt1 = dvae_dev_knob("DVAE_PERS_X3_MT1", 2);
    const int n_rb16 = (N + 15) / 16;
    if ((kind == 2 ? (mt1 & 1) : (mt1 & 2)) && (kind == 2 || kind == 4) && H == 512 &&
        (H / 16) * ((N + 31) / 32) * 2 <= cus && n_rb16 <= 8 && (H / 16) * n_rb16 <= cus)
      mt = 1;
    // forward at H = 512: 8 units x 32 rows per workgroup (lstm_pers_fwd_x3h) where that fits the chip — 64 x 4 workgroups at
    // N = 128 instead of 32 x 4; DVAE_PERS_X3_H8=0 (dev build) keeps the 16-unit kernel
    if (kind == 2 && mt == 2 && dvae_pers_fwd_units(N, H) == 8) units8 = true;
#ifdef DVAE_DEV      // the diagnostics and the sentinel form exist for the 16-unit kernel only
    if (g_pers_dbg || g_pers_nslot > 2 || dvae_dev_knob("DVAE_PERS_SENT", 0)) units8 = false;
#endif
  }
  PersArgs a{};
  a.gates = d.gates; a.wp = (const char*)d.w_packed; a.h_out = (char*)d.h_out; a.c_all = d.c_all;
  a.dh_out = d.dh_out; a.dgates = (char*)d.dgates;
  a.db1 = bwd ? d.dbias_ih : nullptr; a.db2 = bwd ? d.dbias_hh : nullptr;
  a.dbp = bwd ? d.dbias_part : nullptr;
  if (a.dbp) a.db1 = a.db2 = nullptr;
  char* ws = (char*)d.pers_ws;
  a.flags = (unsigned*)ws; a.err = (unsigned*)(ws + PERS_ERR_OFF); a.xch = ws + PERS_XCH_OFF;
  a.T = T; a.N = N; a.ldh = ldh; a.reverse = d.reverse; a.s16 = d.state_bf16 ? 1 : 0;
  a.n_rb = (N + 16 * mt - 1) / (16 * mt);      // (mt = 2 for every fp32x3 / fp32 kernel except the 16-row forms above)
  const unsigned us = d.pers_timeout_us ? d.pers_timeout_us : 2000000u;
  a.timeout = us > 40000000u ? 4000000000u : us * 100u;
  const int64_t slot = pers_slot_bytes(kind, N, H, mt);
  a.xch_bytes = (int)(2 * slot);
#ifdef DVAE_DEV
  a.nslot = 2;
  if (kind == 2) {      // diagnostics of the fp32x3 forward kernel (dvae_lstm_pers_set_dbg)
    a.dbg = g_pers_dbg; a.dbg_frag = g_pers_dbg_frag; a.dbg_jb = g_pers_dbg_jb;
    a.nslot = g_pers_nslot > 2 ? g_pers_nslot : 2;
    a.xch_bytes = (int)(a.nslot * slot);     // (the caller sized the workspace: dvae_lstm_pers_ws_bytes_slots)
  }
#endif
  a.drop_bid = drop_bid;
  // XCD-local hand-off (pers_loc_*): the kernels that have it, where a row group's workgroups share bid % 8; the workgroups
  // verify their placement at frame 1.  DVAE_PERS_XCD_LOCAL=0 in the environment keeps every hand-off write-through.
  static const bool loc_env = []() { const char* e = getenv("DVAE_PERS_XCD_LOCAL"); return !(e && e[0] == '0'); }();
  a.local_ok = (loc_env && (kind == 0 || kind == 1 || (kind == 4 && !units8)) && (a.n_rb % 8) == 0) ? 1 : 0;
#ifdef DVAE_DEV
  a.nodrain = dvae_dev_knob("DVAE_PERS_NODRAIN", 0);
#endif
#ifdef DVAE_PERS_TS
  a.ts = g_pers_ts;
  a.ts_bid = g_pers_ts_bid;
#endif
  const int grid = (kind < 2 ? H / 32 : units8 ? H / 8 : H / 16) * a.n_rb;
  const size_t flag_bytes = kind == 5 ? (size_t)PERS_FLAG_BYTES      // + the partial flags behind the dG flags
                                      : (size_t)a.n_rb * (kind < 2 ? PERS_FLAG_LD : PERS_FLAG_LD_X3) * 4;
#ifdef DVAE_DEV
  static const int sent = dvae_dev_knob("DVAE_PERS_SENT", 0);      // dev build: 1 = the sentinel hand-off form of the fp32x3 forward kernel
  if (kind == 2 && (sent & 1)) {
    // sentinel hand-off: no flags; four slots, the first two poisoned here
    // (the caller sized the workspace with dvae_lstm_pers_ws_bytes: four slots of the largest kind)
    a.xch_bytes = (int)(PERS_NSLOT_S * slot);
    int pb = (int)(2 * slot);
    if (a.nslot > PERS_NSLOT_S) {      // one slot per frame (dvae_lstm_pers_set_dbg): all of them poisoned here
      a.xch_bytes = (int)(a.nslot * slot);
      pb = a.xch_bytes;
    }
    hipLaunchKernelGGL(pers_poison_kernel, dim3(std::min(1024, (pb / 16 + 255) / 256)), dim3(256), 0, s, a.xch, pb);
    if (dvae_check_launch() != DVAE_OK) return DVAE_ELAUNCH;
    return pers_dispatch(12, H, mt, a, grid, s);
  }
#endif
  // (no clearing launch: the flags carry the epoch of their launch, see pers_epoch)
  (void)flag_bytes;
  return pers_dispatch(units8 ? 6 : kind, H, mt, a, grid, s);
}

DVAE_API const unsigned* dvae_lstm_pers_err_word(void* ws) {
  return ws ? (const unsigned*)((char*)ws + PERS_ERR_OFF) : nullptr;
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
  (void)hipMemsetAsync((char*)ws + PERS_ERR_OFF, 0, 16, s);      // reported: clear the sticky record (NOT the flags' epoch behind it)
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

#ifdef DVAE_DEV
// dev build only (scripts/x3_fwd16_diag2.py): register dumps of the fp32x3 forward kernel, ring with `nslot` slots
DVAE_API int dvae_lstm_pers_set_dbg(void* dbg, void* dbg_frag, int dbg_jb, int nslot) {
  g_pers_dbg = (unsigned*)dbg;
  g_pers_dbg_frag = (unsigned*)dbg_frag;
  g_pers_dbg_jb = dbg_jb;
  g_pers_nslot = nslot;
  return DVAE_OK;
}
DVAE_API int64_t dvae_lstm_pers_ws_bytes_slots(int N, int H, int nslot) { return pers_ws_need(nslot, N, H); }
#endif

#ifdef DVAE_PERS_TS
// dev build only (scripts/lstm_pers_timeline.py): stamps of workgroup `bid` go to buf[frame][wave][8]
DVAE_API int dvae_lstm_pers_set_ts(void* buf, int bid) {
  g_pers_ts = (unsigned long long*)buf;
  g_pers_ts_bid = bid;
  return DVAE_OK;
}
#endif
