// fp32 dense contraction on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32).
//
// One kernel serves every GEMM-shaped piece of the training step: Linear fwd/bwd, LSTM input
// projections and weight gradients, and the k=5 Conv1d as an implicit GEMM over frame-major rows
// (a tap is a row shift of +-N*(tap-2), so padding is a plain range check on the row index).
//
// Tile: 128 x (64*NTW) x BK per 256-thread workgroup (4 waves as 2x2, each wave 2 x NTW MFMA tiles of
// 32x32, 32*NTW accumulator registers).  Operands go global -> registers -> LDS, with the next k-tile's
// global loads in flight under the current tile's MFMAs, two LDS buffers, one barrier per k-tile.
// Per-thread source pointers and tap-validity masks are computed once; a k-tile costs one pointer
// increment per load, so the VALU stream between two MFMA blocks stays short.
//
// LDS images (chosen per operand by its memory layout, so staging never transposes):
//   k-contiguous operand ([rows][K] in memory): image [rows][BK + 4 pad]; staged with ds_write_b128,
//     read with BK/8 ds_read_b128 per 32-row MFMA tile per k-tile (row stride 20 or 36 floats: the
//     16-lane groups of a b128 read hit 16 distinct 4-bank slots -> conflict-free).
//   row-contiguous operand ([K][rows] in memory): image [BK][rows + 4 pad]; staged with ds_write_b128
//     along rows, read with ds_read_b32 (32 consecutive floats per half-wave -> conflict-free).
// Both reads use the SAME k order: MFMA step (c, e), e = 0..3, consumes k = 8c + 4*(lane>>5) + e
// on both operands (a permutation of the k order inside the tile, identical for A and B, so the dot
// product is unchanged).  All of a k-tile's fragments are fetched before its MFMAs are issued, so the
// matrix pipe runs back to back instead of paying one LDS round trip per k-step.
#include <cstdlib>
#include <type_traits>
#include <utility>
#include "common.h"
#include "gemm_common.h"

// Development ablations (build with -DDVAE_GEMM_ABL=<bits>, results are WRONG): 1 no operand split (one conversion),
// 2 no global loads inside the k loop, 4 no LDS staging inside the loop, 8 no fragment reads inside the loop, 16 no MFMAs
#ifndef DVAE_GEMM_ABL
#define DVAE_GEMM_ABL 0
#endif
#ifndef DVAE_X3_PD
#define DVAE_X3_PD 2   // k-tiles in flight ahead of the one being computed (split mode), see the kernel
#endif

int g_dvae_compute_mode = 0;   // process default of the contraction mode (DVAE_MODE_*, dvae_set_compute_mode)
int g_dvae_deterministic = 0;  // dvae_set_deterministic: every accumulated output element gets ONE writer in a fixed order

#if defined(DVAE_GEMM_TS) || defined(DVAE_GEMM_TS2)
// Development probe (build with -DDVAE_GEMM_TS): wave 0 of every workgroup measures the s_memtime cycles of its whole
// k-loop (two stamps only: stamps inside the loop serialise it and change what they measure).
__device__ unsigned long long g_gemm_ts[1024 * 8];
#define TS_NOW() __builtin_amdgcn_s_memtime()
#endif
// -DDVAE_GEMM_TS2: the ENDS of the tall kernel — four s_memrealtime stamps (100 MHz, one clock for the whole chip) per
// wave: kernel entry, k loop entered, k loop left, epilogue stores drained (scripts/gemm_ends.py); nothing inside the loop

__device__ __attribute__((aligned(16))) float g_gemm_zero[4] = {0.f, 0.f, 0.f, 0.f};   // what masked lanes load

namespace {

template <bool A_KC, bool B_KC, int NTW, int BK, int WG, int MODE = 0, bool BNS = false, bool A16 = false, bool B16M = false>
__global__ __launch_bounds__(64 * WG * WG) void gemm_f32_kernel(const GemmParams p) {
  constexpr bool BF = (MODE == 1), X3 = (MODE == 2);
  static_assert(!(A16 || B16M) || MODE == 1, "bf16 operands in memory: bf16 mode only");
  constexpr int EA = A16 ? 8 : 4, EB = B16M ? 8 : 4;      // elements per 16-byte load
  constexpr int ESA = A16 ? 2 : 4, ESB = B16M ? 2 : 4;    // bytes per element in memory
  static_assert(!BF || (BK == 32 && WG == 2), "bf16 mode: 128 x 64*NTW x 32 tile only");
  static_assert(!X3 || (BK == 16 && WG == 2), "split mode: 128 x 64*NTW x 16 tile only (3 images x 2 buffers in LDS)");
  constexpr bool B16 = BF || X3;         // bf16 images in LDS
  constexpr int NP = X3 ? 3 : 1;         // images per operand
  constexpr int BM = 64 * WG, NTHR = 64 * WG * WG;
  constexpr int BN = 32 * NTW * WG;      // NTW = 32-wide n-tiles per wave
  constexpr int LD_KC = B16 ? BK + 8 : BK + 4;         // row stride of a k-contiguous image
  constexpr int LDA = A_KC ? LD_KC : (B16 ? BM + 32 : BM + 4);
  constexpr int LDB = B_KC ? LD_KC : (B16 ? BN + 32 : BN + 4);
  constexpr int A_SZ = A_KC ? BM * LD_KC : BK * LDA;
  constexpr int B_SZ = B_KC ? BN * LD_KC : BK * LDB;
  using lds_t = typename std::conditional<B16, __bf16, float>::type;
  constexpr int NLA = BM * BK / EA / NTHR;  // 16-byte loads per thread per k-tile (A)
  constexpr int NLB = BN * BK / EB / NTHR;  // (B)
  constexpr int KQA = BK / EA, KQB = BK / EB;   // 16-byte pieces per row of a k-contiguous tile
  constexpr int NC = BK / 8;                // 8-deep k groups per tile
  __shared__ __attribute__((aligned(16))) lds_t As[2][NP * A_SZ];
  __shared__ __attribute__((aligned(16))) lds_t Bs[2][NP * B_SZ];

  const int t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6;
  const int wm = wave / WG, wn = wave % WG;
  const int l31 = lane & 31, kh = lane >> 5;

  int tile_m, tile_n;
  gemm_tile_of(p, tile_m, tile_n);     // (the 16-wave variant never runs with xcd_map: no registers to spare)
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  int tap_fixed = 0, ks = blockIdx.z;
  if (p.tap_mode == 2) {
    // taps fastest: the five taps of one (tile, k-split) are dispatched back to back AND land on the same XCD (their
    // linear workgroup ids differ by multiples of the tile count, a multiple of 8 for every conv of this model), so
    // they share dY (identical) and X (shifted by four k-tiles) in L2: 408 -> ~200 MB of fabric reads per launch
    tap_fixed = blockIdx.z % p.taps;
    ks = blockIdx.z / p.taps;
  }
  const char* pA = (const char*)p.A;
  const char* pB = (const char*)p.B;
  char* pC = (char*)p.C;
  if (p.batch > 1) {            // batched launch: product b = ks / split_k on its own operands
    const int b = ks / p.split_k;
    ks -= b * p.split_k;
    pA += p.a_boff[b];
    pB += p.b_boff[b];
    pC += p.c_boff[b];
  }
  const int k_begin = ks * p.k_per_split;
  const int k_end = min(p.K, k_begin + p.k_per_split);
  const int klen = k_end - k_begin;
  const int kiters = (klen + BK - 1) / BK;
  const int n_iters = (p.tap_mode == 1 ? p.taps : 1) * kiters;

  // k-split without atomics (p.slab): EVERY split STORES its partial product into its own slab ks (batched: product b has its
  // own run of split_k slabs behind the others'); C is written only by an unsplit launch
  const bool to_slab = p.slab != nullptr && p.split_k > 1;
  if (to_slab) pC = (char*)(p.slab + ((int64_t)(p.batch > 1 ? blockIdx.z / p.split_k : 0) * p.split_k + ks) * p.slab_stride);
  float* __restrict__ C = (float*)pC + (p.tap_mode == 2 ? (int64_t)tap_fixed * p.c_tap_stride : 0);

  // ---- per-thread source descriptors (byte pointers), computed once
  const char* a_src[NLA];
  unsigned a_ok[NLA];   // bit `tap` set: the (shifted) source row exists
  int a_k[NLA];         // this load's k offset inside the tile (bound check against klen)
#pragma unroll
  for (int j = 0; j < NLA; ++j) {
    const int idx = t + NTHR * j;
    if (A_KC) {
      const int row = idx / KQA, kq = idx % KQA;
      const int64_t m = m0 + row;
      a_k[j] = EA * kq;
      a_src[j] = pA + (m * p.lda + k_begin + EA * kq) * ESA;
      unsigned ok = 0;
      if (m < p.M) {
        if (p.tap_mode == 1) {
#pragma unroll
          for (int tp = 0; tp < 5; ++tp) {
            const int64_t ms = m + (int64_t)(tp - 2) * p.a_row_shift;
            ok |= (ms >= 0 && ms < p.M) ? (1u << tp) : 0u;
          }
        } else {
          ok = 1u;
        }
      }
      a_ok[j] = ok;
    } else {
      const int kr = idx / (BM / EA), m4 = idx % (BM / EA);
      a_k[j] = kr;
      a_src[j] = pA + ((int64_t)(k_begin + kr) * p.lda + m0 + EA * m4) * ESA;
      a_ok[j] = (m0 + EA * m4 < p.M) ? 1u : 0u;
    }
  }
  const char* b_src[NLB];
  unsigned b_ok[NLB];
  int b_k[NLB];
#pragma unroll
  for (int j = 0; j < NLB; ++j) {
    const int idx = t + NTHR * j;
    if (B_KC) {
      const int row = idx / KQB, kq = idx % KQB;
      b_k[j] = EB * kq;
      b_src[j] = pB + ((int64_t)(n0 + row) * p.ldb + k_begin + EB * kq) * ESB;
      b_ok[j] = (n0 + row < p.N) ? 1u : 0u;
    } else {
      const int kr = idx / (BN / EB), n4 = idx % (BN / EB);
      b_k[j] = kr;
      const int64_t shift = (p.tap_mode == 2) ? (int64_t)(tap_fixed - 2) * p.bk_row_shift : 0;
      b_src[j] = pB + (((int64_t)(k_begin + kr) + shift) * p.ldb + n0 + EB * n4) * ESB;
      b_ok[j] = (n0 + EB * n4 < p.N) ? 1u : 0u;
    }
  }
  const int64_t b_shift = (p.tap_mode == 2) ? (int64_t)(tap_fixed - 2) * p.bk_row_shift : 0;

  // PD = prefetch distance in k-tiles = number of register sets.  A 16-deep split-mode tile is only 24 MFMAs (768
  // cycles) per wave: one tile of cover is shorter than the L2 / Infinity-Cache latency under load (measured: 3 900
  // cycles per k-tile with PD = 1 against 1 536 of MFMA issue for the two co-resident waves), so tile t+PD is fetched
  // while tile t computes; the sets are small (4 float4 per thread) in this mode.
  constexpr int PD = X3 ? DVAE_X3_PD : 1;
  f32x4 ra_[PD][NLA], rb_[PD][NLB];

  // (tap, kit) of the tile being fetched; uniform
  // Branch-free loads: a lane whose element does not exist (ragged edge, conv padding, k tail) reads 16 bytes of zeros
  // (g_gemm_zero) instead.  With `if (ok) load` hipcc gives every load its own basic block (s_and_saveexec + branch)
  // and a vmcnt(0) at the loop head, and nothing in the load phase can be scheduled next to an MFMA.
  const char* __restrict__ zsrc = (const char*)g_gemm_zero;
  auto load_tiles = [&](f32x4 (&ra)[NLA], f32x4 (&rb)[NLB], int tap, int kit) {
    const int kofs = kit * BK;
    const int64_t a_tap = (p.tap_mode == 1) ? (int64_t)(tap - 2) * p.a_row_shift * p.lda : 0;
    const int64_t b_tap = (p.tap_mode == 1) ? (int64_t)tap * p.b_tap_stride : 0;
#pragma unroll
    for (int j = 0; j < NLA; ++j) {
      bool ok;
      const char* src;
      if (A_KC) {
        ok = ((a_ok[j] >> tap) & 1u) && (kofs + a_k[j] < klen);
        src = a_src[j] + (a_tap + kofs) * ESA;
      } else {
        ok = a_ok[j] && (kofs + a_k[j] < klen);
        src = a_src[j] + (int64_t)kofs * p.lda * ESA;
      }
      ra[j] = *reinterpret_cast<const f32x4*>(ok ? src : zsrc);
    }
#pragma unroll
    for (int j = 0; j < NLB; ++j) {
      bool ok;
      const char* src;
      if (B_KC) {
        ok = b_ok[j] && (kofs + b_k[j] < klen);
        src = b_src[j] + (b_tap + kofs) * ESB;
      } else {
        ok = b_ok[j] && (kofs + b_k[j] < klen);
        if (p.tap_mode == 2) {
          const int64_t kk = (int64_t)k_begin + kofs + b_k[j] + b_shift;
          ok = ok && (kk >= 0) && (kk < p.K);
        }
        src = b_src[j] + (b_tap + (int64_t)kofs * p.ldb) * ESB;
      }
      rb[j] = *reinterpret_cast<const f32x4*>(ok ? src : zsrc);
    }
  };

  auto store_tiles = [&](int buf, f32x4 (&ra)[NLA], f32x4 (&rb)[NLB]) {
#pragma unroll
    for (int j = 0; j < NLA; ++j) {
      const int idx = t + NTHR * j;
      if constexpr (A16) {      // already bf16: 8 elements, one 16-byte LDS store
        const int off = A_KC ? (idx / KQA) * LDA + EA * (idx % KQA) : (idx / (BM / EA)) * LDA + EA * (idx % (BM / EA));
        *reinterpret_cast<f32x4*>(&As[buf][off]) = ra[j];
      } else if constexpr (B16) {
        const int off = A_KC ? (idx / KQA) * LDA + 4 * (idx % KQA) : (idx / (BM / 4)) * LDA + 4 * (idx % (BM / 4));
        if constexpr (X3) {
          bf16x4 pl[3];
          if constexpr (DVAE_GEMM_ABL & 1) pl[0] = pl[1] = pl[2] = __builtin_convertvector(ra[j], bf16x4); else
          split3(ra[j], pl);
#pragma unroll
          for (int q = 0; q < 3; ++q) *reinterpret_cast<bf16x4*>(&As[buf][q * A_SZ + off]) = pl[q];
        } else {
          *reinterpret_cast<bf16x4*>(&As[buf][off]) = __builtin_convertvector(ra[j], bf16x4);
        }
      } else if (A_KC) {
        const int row = idx / KQA, kq = idx % KQA;
        *reinterpret_cast<f32x4*>(&As[buf][row * LDA + 4 * kq]) = ra[j];
      } else {
        const int kr = idx / (BM / 4), m4 = idx % (BM / 4);
        *reinterpret_cast<f32x4*>(&As[buf][kr * LDA + 4 * m4]) = ra[j];
      }
    }
#pragma unroll
    for (int j = 0; j < NLB; ++j) {
      const int idx = t + NTHR * j;
      if constexpr (B16M) {
        const int off = B_KC ? (idx / KQB) * LDB + EB * (idx % KQB) : (idx / (BN / EB)) * LDB + EB * (idx % (BN / EB));
        *reinterpret_cast<f32x4*>(&Bs[buf][off]) = rb[j];
      } else if constexpr (B16) {
        const int off = B_KC ? (idx / KQB) * LDB + 4 * (idx % KQB) : (idx / (BN / 4)) * LDB + 4 * (idx % (BN / 4));
        if constexpr (X3) {
          bf16x4 pl[3];
          if constexpr (DVAE_GEMM_ABL & 1) pl[0] = pl[1] = pl[2] = __builtin_convertvector(rb[j], bf16x4); else
          split3(rb[j], pl);
#pragma unroll
          for (int q = 0; q < 3; ++q) *reinterpret_cast<bf16x4*>(&Bs[buf][q * B_SZ + off]) = pl[q];
        } else {
          *reinterpret_cast<bf16x4*>(&Bs[buf][off]) = __builtin_convertvector(rb[j], bf16x4);
        }
      } else if (B_KC) {
        const int row = idx / KQB, kq = idx % KQB;
        *reinterpret_cast<f32x4*>(&Bs[buf][row * LDB + 4 * kq]) = rb[j];
      } else {
        const int kr = idx / (BN / 4), n4 = idx % (BN / 4);
        *reinterpret_cast<f32x4*>(&Bs[buf][kr * LDB + 4 * n4]) = rb[j];
      }
    }
  };

  // (a 16x16x4 variant of this loop -- finer issue granularity, 155 vs 143-152 TFLOP/s in the register-only probe
  // scripts/mfma_peak.py -- measured the same ~110 TFLOP/s in the full kernel and was dropped)
  f32x16 acc[2][NTW];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NTW; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // Tile order of tap mode 1: taps INNER (k-tile outer).  The five taps of one k-tile read the same 32 columns of five
  // neighbouring row tiles, which the neighbouring workgroups of this XCD read in the same k-tile: ~0.3 MB of
  // activations live per XCD, so each row tile comes over the fabric once.  (Taps outer streams 5 MB per XCD per tap
  // through a 4 MB L2 and every tap misses: measured 199 MB of fabric reads per 512->512 conv launch for 39 MB of
  // operands.)
  const int ntaps_loop = (p.tap_mode == 1) ? p.taps : 1;
  int tap_n = 0, kit_n = 0;   // next tile to fetch
  auto advance = [&]() {
    if constexpr (WG == 4) {   // never a tap mode
      ++kit_n;
      return;
    }
    if (++tap_n == ntaps_loop) {
      tap_n = 0;
      ++kit_n;
    }
  };
  if (n_iters > 0) {
    load_tiles(ra_[0], rb_[0], tap_n, kit_n);
    advance();
    store_tiles(0, ra_[0], rb_[0]);
  }
  if constexpr (PD > 1) {   // tiles 1 .. PD-1 in flight before the loop
#pragma unroll
    for (int u = 1; u < PD; ++u) {
      load_tiles(ra_[u], rb_[u], tap_n, kit_n);
      advance();
    }
  }
  __syncthreads();

  // fragment read bases (constant across k-tiles)
  const int a_frag = A_KC ? (wm * 64 + l31) * LDA + 4 * kh : (4 * kh) * LDA + wm * 64 + l31;
  const int b_frag = B_KC ? (wn * 32 * NTW + l31) * LDB + 4 * kh : (4 * kh) * LDB + wn * 32 * NTW + l31;

  int cur = 0;
#ifdef DVAE_GEMM_TS
  const unsigned long long ts_t0 = TS_NOW(), ts_r0 = __builtin_amdgcn_s_memrealtime();
#endif
  auto compute_tile = [&]() {
    if constexpr (B16) {
      // v_mfma_f32_32x32x16_bf16: lane (r = lane&31, h = lane>>5) holds k = 16s + 8h + j, j = 0..7, of row r.
      // k-contiguous image: one ds_read_b128.  Row-contiguous image [k][rows]: two ds_read_b64_tr_b16; lane 4q+pq of
      // the 16-lane group g supplies &img[k0 + q][r0 + 4pq] (k0 = 16s + 8(g>>1) (+4), r0 = 16(g&1)) and receives
      // 4 consecutive k of row r0 + (lane&15).
      const int g16 = lane >> 4, li = lane & 15;
      const int tr_k = 8 * (g16 >> 1) + (li >> 2), tr_r = 16 * (g16 & 1) + 4 * (li & 3);
      typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
      auto frag = [&](const lds_t* img, bool kc, int ld, int row0, int s2) -> bf16x8 {
        if (kc) return *reinterpret_cast<const bf16x8*>(&img[(row0 + l31) * ld + 16 * s2 + 8 * kh]);
        const lds_t* q0 = &img[(16 * s2 + tr_k) * ld + row0 + tr_r];
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(q0));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(q0 + 4 * ld));
        return __builtin_shufflevector(__builtin_bit_cast(bf16x4, lo), __builtin_bit_cast(bf16x4, hi), 0, 1, 2, 3, 4, 5, 6, 7);
      };
      constexpr int NS = BK / 16;
      bf16x8 av[NS][2][NP], bv[NS][NTW][NP];
#pragma unroll
      for (int s2 = 0; s2 < NS; ++s2)
#pragma unroll
        for (int q = 0; q < NP; ++q) {
#pragma unroll
          for (int mt = 0; mt < 2; ++mt) {
            if constexpr (DVAE_GEMM_ABL & 8) av[s2][mt][q] = __builtin_bit_cast(bf16x8, ra_[0][0]);
            else av[s2][mt][q] = frag(&As[cur][q * A_SZ], A_KC, LDA, wm * 64 + mt * 32, s2);
          }
#pragma unroll
          for (int nt = 0; nt < NTW; ++nt) {
            if constexpr (DVAE_GEMM_ABL & 8) bv[s2][nt][q] = __builtin_bit_cast(bf16x8, rb_[0][0]);
            else bv[s2][nt][q] = frag(&Bs[cur][q * B_SZ], B_KC, LDB, wn * 32 * NTW + nt * 32, s2);
          }
        }
      __builtin_amdgcn_sched_barrier(0);
      // X3: the six partial products of weight >= 2^-16, (a1,b1) first: it needs only the first image of each operand
      constexpr int NT6 = X3 ? 6 : 1;
      constexpr int ia[6] = {0, 0, 1, 1, 0, 2}, ib[6] = {0, 1, 0, 1, 2, 0};
#pragma unroll
      for (int s2 = 0; s2 < NS; ++s2)
#pragma unroll
        for (int term = 0; term < NT6; ++term)
#pragma unroll
          for (int nt = 0; nt < NTW; ++nt) {
            if constexpr (DVAE_GEMM_ABL & 16) {
              asm volatile("" ::"v"(av[s2][0][ia[term]]), "v"(av[s2][1][ia[term]]), "v"(bv[s2][nt][ib[term]]));
            } else {
            acc[0][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[s2][0][ia[term]], bv[s2][nt][ib[term]], acc[0][nt], 0, 0, 0);
            acc[1][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[s2][1][ia[term]], bv[s2][nt][ib[term]], acc[1][nt], 0, 0, 0);
            }
          }
      __builtin_amdgcn_sched_barrier(0);
    } else {
    // ---- fragments in 8-deep k groups: group c, element e <-> k = 8c + 4*kh + e.  The reads of group c+1 are
    // issued BEFORE the 4*2*NTW MFMAs of group c (two register sets, order pinned with sched_barrier), so the LDS
    // round trip runs under the matrix pipe instead of between MFMA bursts (measured: the burst structure hipcc
    // picks on its own leaves the pipe idle ~20 % of the time even without any barrier).
    auto read_group = [&](int c, float (&av)[2][4], float (&bv)[NTW][4]) {
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        if (A_KC) {
          const f32x4 v = *reinterpret_cast<const f32x4*>(&As[cur][a_frag + mt * 32 * LDA + 8 * c]);
#pragma unroll
          for (int e = 0; e < 4; ++e) av[mt][e] = v[e];
        } else {
          const float* src = &As[cur][a_frag + mt * 32];
#pragma unroll
          for (int e = 0; e < 4; ++e) av[mt][e] = src[(8 * c + e) * LDA];
        }
      }
#pragma unroll
      for (int nt = 0; nt < NTW; ++nt) {
        if (B_KC) {
          const f32x4 v = *reinterpret_cast<const f32x4*>(&Bs[cur][b_frag + nt * 32 * LDB + 8 * c]);
#pragma unroll
          for (int e = 0; e < 4; ++e) bv[nt][e] = v[e];
        } else {
          const float* src = &Bs[cur][b_frag + nt * 32];
#pragma unroll
          for (int e = 0; e < 4; ++e) bv[nt][e] = src[(8 * c + e) * LDB];
        }
      }
    };
    auto mfma_group = [&](float (&av)[2][4], float (&bv)[NTW][4]) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) {
          acc[0][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[0][e], bv[nt][e], acc[0][nt], 0, 0, 0);
          acc[1][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[1][e], bv[nt][e], acc[1][nt], 0, 0, 0);
        }
      }
    };
    float a0[2][4], b0[NTW][4], a1[2][4], b1[NTW][4];
    read_group(0, a0, b0);
#pragma unroll
    for (int c = 0; c < NC; c += 2) {
      if (c + 1 < NC) read_group(c + 1, a1, b1);
      __builtin_amdgcn_sched_barrier(0);
      mfma_group(a0, b0);
      __builtin_amdgcn_sched_barrier(0);
      if (c + 2 < NC) read_group(c + 2, a0, b0);
      __builtin_amdgcn_sched_barrier(0);
      if (c + 1 < NC) mfma_group(a1, b1);
      __builtin_amdgcn_sched_barrier(0);
    }
    }
  };
  if constexpr (PD == 1) {
    for (int it = 0; it < n_iters; ++it) {
      const bool more = (it + 1 < n_iters);
      if (more) {
        load_tiles(ra_[0], rb_[0], tap_n, kit_n);
        advance();
      }
      compute_tile();
      if (more) store_tiles(cur ^ 1, ra_[0], rb_[0]);
      __syncthreads();
      cur ^= 1;
    }
  } else {
    // tile t lives in set t % PD: fetched at the top of iteration t - PD, staged into LDS at the end of iteration t - 1.
    // The loop body is branch-free: the iteration count is rounded up to a multiple of PD and tiles past the end load
    // zeros (the k bound check of load_tiles) — at most PD-1 wasted tiles per workgroup — so the PD iterations of a trip
    // are one basic block: the accumulators stay in place and the compiler keeps COUNTED vmcnt waits (the stage of tile
    // it+1 waits for its own loads only, the younger sets stay in flight across the barrier).
    for (int it0 = 0; it0 < n_iters; it0 += PD) {
#pragma unroll
      for (int u = 0; u < PD; ++u) {
        // set u is free (tile it0+u was staged an iteration ago): tile it0 + u + PD goes there
        if constexpr (!(DVAE_GEMM_ABL & 2)) load_tiles(ra_[u], rb_[u], tap_n, kit_n);
        advance();
        compute_tile();
        if constexpr (!(DVAE_GEMM_ABL & 4)) store_tiles(cur ^ 1, ra_[(u + 1) % PD], rb_[(u + 1) % PD]);
        __syncthreads();
        if constexpr (!(DVAE_GEMM_ABL & 4)) cur ^= 1;
      }
    }
  }
#ifdef DVAE_GEMM_TS
  if (threadIdx.x == 0 && blockIdx.x < 1024 && blockIdx.z == 0) {
    unsigned long long* o = g_gemm_ts + blockIdx.x * 8;
    o[0] = TS_NOW() - ts_t0;
    o[1] = (unsigned long long)n_iters;
    o[2] = __builtin_amdgcn_s_memrealtime() - ts_r0;   // 100 MHz reference: o[0] / o[2] x 100 MHz = the clock held
  }
#endif

  // ---- epilogue: C/D lane map of the 32x32 tile: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
  const bool add_bias = (p.bias != nullptr) && (ks == 0);
  const int epi = to_slab ? DVAE_EPI_STORE : p.epi, act = p.act;
  // BNS: BatchNorm partial statistics of this wave's 64 rows x 32*NTW columns (= one 64-row chunk of bn.hip), gathered
  // in the SAME pass that stores the tile (each accumulator is read once): sum and sum of squares per statistics group,
  // fp32 over the 64 values of a chunk, fp64 from there on (bn_stats_finalize).  A separate instantiation: in the
  // common one this code cost 120 VGPRs (occupancy 1 instead of 2-3).
  float bst[NTW][4];    // [column tile][group 0: sum, sumsq; group 1: sum, sumsq]
  int bmod = 0;
  if constexpr (BNS) {
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
      for (int q = 0; q < 4; ++q) bst[nt][q] = 0.f;
    bmod = (m0 + wm * 64 + 4 * kh) % p.bn_nseg;
  }
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {
      const int col = n0 + wn * 32 * NTW + nt * 32 + l31;
      if (col >= p.N) continue;
      const float bias_v = add_bias ? p.bias[col] : 0.f;
      const int row0 = m0 + wm * 64 + mt * 32 + 4 * kh;
      float* cbase = C + (int64_t)row0 * p.ldc + col;
      if constexpr (!BNS) {
        if (p.c_vec && epi != DVAE_EPI_ATOMIC) {
          // 16-byte stores (as in the 256 x 128 kernels): the accumulator holds 4 consecutive ROWS of one column per lane; a
          // 4 x 4 transpose inside each quad of lanes (two DPP butterfly stages) turns them into 4 consecutive COLUMNS of one
          // row.  64 single-dword stores per lane made the short-K launches (K = 128 projections and outer products: 8
          // k-steps in front of a 64 KB tile) store-issue-bound.  Same values, wider stores: results are bit-identical.
          const int q4 = lane & 3;
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            float x[4], y[4], z[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) x[e] = act_apply(acc[mt][nt][4 * g + e] + bias_v, act);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float o = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x[e ^ 1]), 0xB1, 0xF, 0xF, true));
              y[e] = ((q4 ^ e) & 1) ? o : x[e];
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float o = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, y[e ^ 2]), 0x4E, 0xF, 0xF, true));
              z[e] = ((q4 ^ e) & 2) ? o : y[e];
            }
            // this lane now holds row (8 g + 4 kh + q4) of the tile, columns 4 (l31 >> 2) .. + 3
            const int row = row0 + 8 * g + q4;
            if (row < p.M) {
              f32x4* dst = reinterpret_cast<f32x4*>(C + (int64_t)row * p.ldc + (col - q4));
              f32x4 o4 = {z[0], z[1], z[2], z[3]};
              if (epi == DVAE_EPI_ACCUM) o4 += *dst;
              *dst = o4;
            }
          }
          continue;
        }
      }
      if (epi == DVAE_EPI_STORE && p.c16) {       // bf16 output (bf16 mode: the consumer is another contraction)
        __bf16* cb = (__bf16*)pC + (int64_t)row0 * p.ldc + col;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int dr = (r & 3) + 8 * (r >> 2);
          if (row0 + dr < p.M) cb[(int64_t)dr * p.ldc] = (__bf16)act_apply(acc[mt][nt][r] + bias_v, act);
        }
      } else if (epi == DVAE_EPI_STORE) {
        if (act == DVAE_ACT_NONE) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int dr = (r & 3) + 8 * (r >> 2);
            const float v = acc[mt][nt][r] + bias_v;
            const bool ok = row0 + dr < p.M;
            if (ok) cbase[(int64_t)dr * p.ldc] = v;
            if constexpr (BNS) {
              const int nseg = p.bn_nseg;
              int rm = bmod + mt * 32 + dr;
              if (nseg >= 64) rm -= (rm >= nseg) ? nseg : 0; else rm %= nseg;
              const float u = ok ? v : 0.f;
              const bool g1 = rm >= nseg / p.bn_groups;
              bst[nt][0] += g1 ? 0.f : u;
              bst[nt][1] += g1 ? 0.f : u * u;
              bst[nt][2] += g1 ? u : 0.f;
              bst[nt][3] += g1 ? u * u : 0.f;
            }
          }
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int dr = (r & 3) + 8 * (r >> 2);
            if (row0 + dr < p.M) cbase[(int64_t)dr * p.ldc] = act_apply(acc[mt][nt][r] + bias_v, act);
          }
        }
      } else if (epi == DVAE_EPI_ACCUM) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int dr = (r & 3) + 8 * (r >> 2);
          if (row0 + dr < p.M) cbase[(int64_t)dr * p.ldc] += acc[mt][nt][r] + bias_v;
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int dr = (r & 3) + 8 * (r >> 2);
          if (row0 + dr < p.M) atomicAdd(cbase + (int64_t)dr * p.ldc, acc[mt][nt][r] + bias_v);
        }
      }
    }
  }
  if constexpr (BNS) {
    const int chunk = tile_m * 2 + wm, nchunks = (p.M + DVAE_BN_ROWS_PER_CHUNK - 1) / DVAE_BN_ROWS_PER_CHUNK;
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {
      const int col = n0 + wn * 32 * NTW + nt * 32 + l31;
#pragma unroll
      for (int q = 0; q < 4; ++q) bst[nt][q] += __shfl_xor(bst[nt][q], 32, 64);     // the two lane halves of a column
      if (kh == 0 && col < p.N && chunk < nchunks) {
        double* o = p.bn_part + ((int64_t)chunk * p.bn_groups * p.N + col) * 2;
        o[0] = (double)bst[nt][0];
        o[1] = (double)bst[nt][1];
        if (p.bn_groups > 1) {
          o[(int64_t)p.N * 2] = (double)bst[nt][2];
          o[(int64_t)p.N * 2 + 1] = (double)bst[nt][3];
        }
      }
    }
  }
}


// ------------------------------------------------------------------------------------------------------------------
// Split mode, 256 x 128 x 16 tile ("tall"): 4 waves (2 x 2), each a 128 x 64 output tile = 4 x 2 MFMA tiles, ONE
// workgroup per CU, one wave per SIMD.  Why this shape (measured, scripts/coissue.py, gemm_ablate.py, mfma_peak.py):
//  * the split arithmetic of one wave and the MFMAs of ANOTHER wave of the same SIMD do not overlap at all (776 + 508
//    cycles alone, 1284 together): two workgroups per CU (the 128 x 128 kernel) or two alternating wave groups buy no
//    overlap for VALU-heavy staging — only instructions of the wave's OWN stream placed in the 32-cycle shadow of its
//    MFMAs (<= ~5 four-cycle issue slots per MFMA, MI355X_MICROARCH.md) are free;
//  * a 128 x 64 wave tile has 48 MFMAs per k-step to hide under: ~240 issue slots against ~190 instructions of staging
//    (split of 24 values, 9 ds_write_b128, 18 ds_read_b128, 6 buffer loads), and the 256 x 128 tile moves 3/4 of the
//    L2 -> CU bytes of two 128 x 128 tiles (whose data movement alone runs at the L2 bandwidth limit).
// So the k loop is ONE basic block, software-pipelined four k-tiles deep, with the interleave pinned by
// sched_group_barrier:
//     iteration i:   buffer loads of tile i+3 -> register set (i+1)%2       (out-of-range rows / conv padding / ragged
//                    fragment reads of tile i+1 (LDS buffer (i+1)%2)         edges read as zeros: raw buffer bounds check,
//                    48 MFMAs on fragment set i%2, the split + LDS writes     no per-load select)
//                    of tile i+2 (register set i%2 -> LDS buffer i%2) in their shadow
//                    ONE barrier
// Tiles past the end read zeros (offsets past the operand), the iteration count is rounded up to even.
// Requires: every k range a multiple of 16 and operands below 2 GiB (launch_gemm falls back to the 128 x 128 kernel).
// BNU (with BNS): every 64-row statistics chunk lies inside ONE group (group size a multiple of 64): two FMAs per element
// instead of the per-element row arithmetic and four selects; a separate instantiation (as a run-time branch inside the
// unrolled epilogue it made both paths slower)
template <bool A_KC, bool B_KC, bool BNS, bool BNU = false>
__global__ __launch_bounds__(256) void gemm_x3_tall_kernel(const GemmParams p) {
  constexpr int BM = 256, BN = 128, BK = 16, NTHR = 256, MTW = 4, NTW = 2;
  constexpr int LD_KC = BK + 8;
  constexpr int LDA = A_KC ? LD_KC : BM + 32;
  constexpr int LDB = B_KC ? LD_KC : BN + 32;
  constexpr int A_SZ = A_KC ? BM * LD_KC : BK * LDA;
  constexpr int B_SZ = B_KC ? BN * LD_KC : BK * LDB;
  constexpr int NUA = BM * BK / 8 / NTHR, NUB = BN * BK / 8 / NTHR;   // 8-element units per thread per k-tile: 2, 1
  constexpr int NU = NUA + NUB;
  // an offset no operand reaches (operands < 1 GiB, launch_gemm): reads zeros.  Sums stay outside: a real offset or a
  // small negative tap offset plus OOB, and OOB + OOB = 0x80000000
  constexpr unsigned OOB = 0xC0000000u;
  __shared__ __attribute__((aligned(16))) __bf16 As[2][3 * A_SZ];
  __shared__ __attribute__((aligned(16))) __bf16 Bs[2][3 * B_SZ];
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

  const int t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, kh = lane >> 5;
#ifdef DVAE_GEMM_TS2
  const unsigned long long t2_entry = __builtin_amdgcn_s_memrealtime();
#endif

  int tile_m, tile_n;
  gemm_tile_of(p, tile_m, tile_n);
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  int tap_fixed = 0, ks = blockIdx.z;
  if (p.tap_mode == 2) {
    tap_fixed = blockIdx.z % p.taps;
    ks = blockIdx.z / p.taps;
  }
  const int k_begin = ks * p.k_per_split;
  const int k_end = min(p.K, k_begin + p.k_per_split);
  const int kiters = (k_end - k_begin) / BK;
  const int n_iters = (p.tap_mode == 1 ? p.taps : 1) * kiters;
  const bool to_slab = p.slab != nullptr && p.split_k > 1;      // k-split without atomics: see gemm_f32_kernel
  float* __restrict__ C = (to_slab ? p.slab + (int64_t)ks * p.slab_stride : (float*)p.C) +
                          (p.tap_mode == 2 ? (int64_t)tap_fixed * p.c_tap_stride : 0);

  // ---- operands as raw buffers: byte offsets, unsigned 32-bit; anything outside [0, bytes) reads zeros.  A row shifted
  // outside the matrix by a conv tap, a row past M / N, a k-row outside [0, K) (wgrad taps) or a tile past the end all
  // land outside; what wraps INSIDE instead (columns past the edge of a row-contiguous operand) is masked per thread.
  const int a_rows = A_KC ? p.M : p.K, b_rows = B_KC ? p.N : p.K;
  const __amdgpu_buffer_rsrc_t a_rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, (int)((int64_t)a_rows * p.lda * 4), 0x00020000);
  const int64_t b_bytes = (int64_t)b_rows * p.ldb * 4 * ((p.tap_mode == 1) ? p.taps : 1);
  const __amdgpu_buffer_rsrc_t b_rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.B, 0, (int)b_bytes, 0x00020000);
  unsigned a_vo[NUA], b_vo[NUB];   // byte offset of the unit's first element in k-tile 0 (tap 2)
  int a_off[NUA], b_off[NUB];      // LDS element offset of the unit
#pragma unroll
  for (int j = 0; j < NUA; ++j) {
    const int idx = t + NTHR * j;
    if (A_KC) {
      const int row = idx >> 1, k8 = (idx & 1) * 8;
      a_off[j] = row * LDA + k8;
      a_vo[j] = (unsigned)(((int64_t)(m0 + row) * p.lda + k_begin + k8) * 4);
      if (m0 + row >= p.M) a_vo[j] = OOB;
    } else {
      const int kr = idx / (BM / 8), m8 = (idx % (BM / 8)) * 8;
      a_off[j] = kr * LDA + m8;
      a_vo[j] = (unsigned)(((int64_t)(k_begin + kr) * p.lda + m0 + m8) * 4);
      if (m0 + m8 >= p.M) a_vo[j] = OOB;
    }
  }
  const int64_t b_shift = (p.tap_mode == 2) ? (int64_t)(tap_fixed - 2) * p.bk_row_shift : 0;
#pragma unroll
  for (int j = 0; j < NUB; ++j) {
    const int idx = t + NTHR * j;
    if (B_KC) {
      const int row = idx >> 1, k8 = (idx & 1) * 8;
      b_off[j] = row * LDB + k8;
      b_vo[j] = (unsigned)(((int64_t)(n0 + row) * p.ldb + k_begin + k8) * 4);
      if (n0 + row >= p.N) b_vo[j] = OOB;
    } else {
      const int kr = idx / (BN / 8), n8 = (idx % (BN / 8)) * 8;
      b_off[j] = kr * LDB + n8;
      b_vo[j] = (unsigned)(((int64_t)(k_begin + kr) + b_shift) * p.ldb * 4 + (int64_t)(n0 + n8) * 4);   // may be "negative": outside
      if (n0 + n8 >= p.N) b_vo[j] = OOB;
    }
  }
  // per-tile scalar offsets (uniform): tap part + k part
  const int ntaps_loop = (p.tap_mode == 1) ? p.taps : 1;
  const int a_tap_step = (p.tap_mode == 1) ? (int)(p.a_row_shift * p.lda * 4) : 0;
  const int b_tap_step = (p.tap_mode == 1) ? (int)(p.b_tap_stride * 4) : 0;
  const int a_k_step = A_KC ? BK * 4 : (int)(BK * p.lda * 4);
  const int b_k_step = B_KC ? BK * 4 : (int)(BK * p.ldb * 4);
  int tap_n = 0, kit_n = 0;   // next tile to fetch
  struct Regs {
    u32x4 a[NUA][2], b[NUB][2];
  };
  auto load_tiles = [&](Regs& g) {
    // past the last k-tile: an offset outside every operand
    const bool live = kit_n < kiters;
    const unsigned a_s = live ? (unsigned)((tap_n - 2) * a_tap_step + kit_n * a_k_step) : OOB;
    const unsigned b_s = live ? (unsigned)(tap_n * b_tap_step + kit_n * b_k_step) : OOB;
#pragma unroll
    for (int j = 0; j < NUA; ++j) {
      g.a[j][0] = __builtin_amdgcn_raw_buffer_load_b128(a_rs, a_vo[j] + a_s, 0, 0);
      g.a[j][1] = __builtin_amdgcn_raw_buffer_load_b128(a_rs, a_vo[j] + a_s, 16, 0);
    }
#pragma unroll
    for (int j = 0; j < NUB; ++j) {
      g.b[j][0] = __builtin_amdgcn_raw_buffer_load_b128(b_rs, b_vo[j] + b_s, 0, 0);
      g.b[j][1] = __builtin_amdgcn_raw_buffer_load_b128(b_rs, b_vo[j] + b_s, 16, 0);
    }
    const bool wrap = (tap_n + 1 == ntaps_loop);
    tap_n = wrap ? 0 : tap_n + 1;
    kit_n += wrap ? 1 : 0;
  };
  // unit (8 values) -> three bf16x8, one ds_write_b128 per plane
  auto stage_unit = [&](__bf16* img, int sz, const u32x4 (&v)[2]) {
    bf16x4 lo[3], hi[3];
    split3(__builtin_bit_cast(f32x4, v[0]), lo);
    split3(__builtin_bit_cast(f32x4, v[1]), hi);
#pragma unroll
    for (int q = 0; q < 3; ++q)
      *reinterpret_cast<bf16x8*>(&img[q * sz]) = __builtin_shufflevector(lo[q], hi[q], 0, 1, 2, 3, 4, 5, 6, 7);
  };
  auto stage_all = [&](int buf, Regs& g) {
#pragma unroll
    for (int j = 0; j < NUA; ++j) stage_unit(&As[buf][a_off[j]], A_SZ, g.a[j]);
#pragma unroll
    for (int j = 0; j < NUB; ++j) stage_unit(&Bs[buf][b_off[j]], B_SZ, g.b[j]);
  };

  // fragments of v_mfma_f32_32x32x16_bf16 (see compute_tile of the kernel above)
  const int g16 = lane >> 4, li = lane & 15;
  const int tr_k = 8 * (g16 >> 1) + (li >> 2), tr_r = 16 * (g16 & 1) + 4 * (li & 3);
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  auto frag = [&](const __bf16* img, bool kc, int ld, int row0) -> bf16x8 {
    if (kc) return *reinterpret_cast<const bf16x8*>(&img[(row0 + l31) * ld + 8 * kh]);
    const __bf16* q0 = &img[tr_k * ld + row0 + tr_r];
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(q0));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(q0 + 4 * ld));
    return __builtin_shufflevector(__builtin_bit_cast(bf16x4, lo), __builtin_bit_cast(bf16x4, hi), 0, 1, 2, 3, 4, 5, 6, 7);
  };
  struct Frags {
    bf16x8 a[MTW][3], b[NTW][3];
  };
  auto read_frags = [&](Frags& f, int buf) {
#pragma unroll
    for (int q = 0; q < 3; ++q) {
#pragma unroll
      for (int mt = 0; mt < MTW; ++mt) f.a[mt][q] = frag(&As[buf][q * A_SZ], A_KC, LDA, wm * 128 + mt * 32);
#pragma unroll
      for (int nt = 0; nt < NTW; ++nt) f.b[nt][q] = frag(&Bs[buf][q * B_SZ], B_KC, LDB, wn * 64 + nt * 32);
    }
  };

  f32x16 acc[MTW][NTW];
#pragma unroll
  for (int i = 0; i < MTW; ++i)
#pragma unroll
    for (int j = 0; j < NTW; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  Regs gs[2];
  Frags fr[2];
  // ---- prologue: tile 0 staged and its fragments read, tile 1 staged, tile 2 in flight
  load_tiles(gs[0]);
  load_tiles(gs[1]);
  stage_all(0, gs[0]);
  load_tiles(gs[0]);
  __syncthreads();
  read_frags(fr[0], 0);
  stage_all(1, gs[1]);
  __syncthreads();

  // ---- the k loop as 48 pinned steps per k-tile: step g = MFMA g, then AT MOST five other instructions (what fits in
  // the MFMA's 32-cycle shadow); sched_barrier(0) after every step keeps hipcc from clumping them (left alone it puts
  // 37 MFMAs back to back and the split arithmetic behind them; sched_group_barrier pipelines came out no better).
  //   steps 0-17   one fragment read of tile i+1 each; steps 0-5 also one buffer load of tile i+3
  //   steps 18-46  the split of tile i+2, one PAIR-LEVEL per step (v_cvt_pk_bf16_f32, shift, mask, two subtracts: 5
  //                instructions), a unit's three ds_write_b128 riding on its last step and the two after it
  constexpr int ia[6] = {0, 0, 1, 1, 0, 2}, ib[6] = {0, 1, 0, 1, 2, 0};
  unsigned a_s = 0, b_s = 0;    // scalar byte offsets of the tile being fetched (uniform)
  unsigned pl[2][3][4];         // planes of the unit being split (4 pairs each); two sets: a unit's last two writes
  float rr[2][8];               // overlap the next unit's first levels.  rr: the residuals
  auto unit_regs = [&](Regs& g, int j) -> const u32x4(&)[2] { return j < NUA ? g.a[j] : g.b[j - NUA]; };
  auto fsub = [](float a, float b) {   // (asm: a plain a - b gets SLP-packed into v_pk_add_f32, which costs ~13 extra cycles beside an MFMA)
    float r;
    asm("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
  };
  auto step = [&](auto G, auto U) {
    constexpr int g = decltype(G)::value, u = decltype(U)::value;
    {
      constexpr int term = g / 8, mt = (g % 8) / 2, nt = g % 2;
      acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[u].a[mt][ia[term]], fr[u].b[nt][ib[term]], acc[mt][nt], 0, 0, 0);
    }
    // loads of tile i+3 -> set u^1 (steps 0-5); fragments of tile i+1 <- LDS buffer u^1, ONE read per step (steps 0-17):
    // the four waves of the CU share one LDS pipe (a ds_read_b128 of a wave occupies it for 8 cycles) and pass the barrier
    // together — three reads per step made each of them queue behind the other three
    if constexpr (g < 6) {
      Regs& gn = gs[u ^ 1];
      constexpr int j = g / 2, hv = g % 2;
      if constexpr (j < NUA) gn.a[j][hv] = __builtin_amdgcn_raw_buffer_load_b128(a_rs, a_vo[j] + a_s, 16 * hv, 0);
      else gn.b[j - NUA][hv] = __builtin_amdgcn_raw_buffer_load_b128(b_rs, b_vo[j - NUA] + b_s, 16 * hv, 0);
    }
    if constexpr (g < 18) {
      Frags& f = fr[u ^ 1];
      constexpr int q = g / 6, w = g % 6;            // plane-major, A tiles then B tiles
      if constexpr (w < MTW) f.a[w][q] = frag(&As[u ^ 1][q * A_SZ], A_KC, LDA, wm * 128 + w * 32);
      else f.b[w - MTW][q] = frag(&Bs[u ^ 1][q * B_SZ], B_KC, LDB, wn * 64 + (w - MTW) * 32);
    } else {
      // the split of tile i+2 (set u -> LDS buffer u): unit j = steps 18 + 9 j .. 18 + 9 j + 8 (eight pair-levels and the
      // last conversion); its three ds_write_b128 ride on its last step and the two steps after it
      constexpr int s = g - 18;             // 0..29
      constexpr int SPU = 9;
      constexpr int j = s / SPU, w = s % SPU;
      if constexpr (j < NU) {
        const u32x4(&v)[2] = unit_regs(gs[u], j);
        if constexpr (w < 8) {
          constexpr int lvl = w / 4, i = w % 4;
          if constexpr (lvl == 0) {   // (bit_cast of the whole vector: on a vector ELEMENT lvalue it reads element 0)
            const f32x4 xv = __builtin_bit_cast(f32x4, v[i / 2]);
            rr[j & 1][2 * i] = xv[2 * (i % 2)];
            rr[j & 1][2 * i + 1] = xv[2 * (i % 2) + 1];
          }
          const f32x2 xy = {rr[j & 1][2 * i], rr[j & 1][2 * i + 1]};
          const unsigned pk = __builtin_bit_cast(unsigned, __builtin_convertvector(xy, bf16x2));
          pl[j & 1][lvl][i] = pk;
          rr[j & 1][2 * i] = fsub(rr[j & 1][2 * i], __builtin_bit_cast(float, pk << 16));
          rr[j & 1][2 * i + 1] = fsub(rr[j & 1][2 * i + 1], __builtin_bit_cast(float, pk & 0xffff0000u));
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const f32x2 xy = {rr[j & 1][2 * i], rr[j & 1][2 * i + 1]};
            pl[j & 1][2][i] = __builtin_bit_cast(unsigned, __builtin_convertvector(xy, bf16x2));
          }
        }
      }
      // writes of unit jw: plane 0 on the unit's last step, planes 1 and 2 on the next two steps
      constexpr int sw = s - 8;             // >= 0 from the first unit's last step on
      if constexpr (sw >= 0 && sw % SPU < 3 && sw / SPU < NU) {
        constexpr int jw = sw / SPU, q = sw % SPU;
        __bf16* img = jw < NUA ? &As[u][a_off[jw < NUA ? jw : 0]] : &Bs[u][b_off[jw < NUA ? 0 : jw - NUA]];
        constexpr int sz = jw < NUA ? A_SZ : B_SZ;
        const u32x4 o = {pl[jw & 1][q][0], pl[jw & 1][q][1], pl[jw & 1][q][2], pl[jw & 1][q][3]};
        *reinterpret_cast<u32x4*>(&img[q * sz]) = o;
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  };
#ifdef DVAE_GEMM_TS
  unsigned long long ts_a = 0, ts_b = 0, ts_c = 0, ts_bar = 0, ts_x = TS_NOW();
  const unsigned long long ts_t0 = ts_x, ts_r0 = __builtin_amdgcn_s_memrealtime();
#define TALL_LAP(acc_) do { const unsigned long long n_ = TS_NOW(); acc_ += n_ - ts_x; ts_x = n_; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define TALL_LAP(acc_) do { } while (0)
#endif
  auto all_steps = [&](auto U) {
    for_seq([&](auto G) {
      step(G, U);
      if constexpr (decltype(G)::value == 17) TALL_LAP(ts_a);
      if constexpr (decltype(G)::value == 35) TALL_LAP(ts_b);
      if constexpr (decltype(G)::value == 47) TALL_LAP(ts_c);
    }, std::make_integer_sequence<int, 48>{});
  };
  auto next_tile_offsets = [&]() {
    const bool live = kit_n < kiters;
    a_s = live ? (unsigned)((tap_n - 2) * a_tap_step + kit_n * a_k_step) : OOB;
    b_s = live ? (unsigned)(tap_n * b_tap_step + kit_n * b_k_step) : OOB;
    const bool wrap = (tap_n + 1 == ntaps_loop);
    tap_n = wrap ? 0 : tap_n + 1;
    kit_n += wrap ? 1 : 0;
  };
#ifdef DVAE_GEMM_TS2
  const unsigned long long t2_loop0 = __builtin_amdgcn_s_memrealtime();
#endif
  for (int it0 = 0; it0 < n_iters; it0 += 2) {
    next_tile_offsets();
    all_steps(std::integral_constant<int, 0>{});
    __syncthreads();
    TALL_LAP(ts_bar);
    next_tile_offsets();
    all_steps(std::integral_constant<int, 1>{});
    __syncthreads();
    TALL_LAP(ts_bar);
  }
#ifdef DVAE_GEMM_TS2
  const unsigned long long t2_loop1 = __builtin_amdgcn_s_memrealtime();
#endif
#ifdef DVAE_GEMM_TS
  if ((threadIdx.x & 63) == 0 && blockIdx.x < 256 && blockIdx.z == 0) {
    unsigned long long* o = g_gemm_ts + (blockIdx.x * 4 + wave) * 8;
    o[0] = TS_NOW() - ts_t0;
    o[1] = (unsigned long long)n_iters;
    o[2] = __builtin_amdgcn_s_memrealtime() - ts_r0;
    o[3] = ts_a; o[4] = ts_b; o[5] = ts_c; o[6] = ts_bar;
  }
#endif

  // ---- epilogue: C/D lane map of the 32x32 tile: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5).
  // One straight-line copy per (epilogue kind, activation): with `epi` / `act` tested per element the 128 values of a
  // lane went through ~1 000 scalar branches — 37 k cycles per tile (in-kernel stamps), a quarter of a 64-step k loop.
  const bool add_bias = (p.bias != nullptr) && (ks == 0);
  auto emit = [&](auto EPI_, auto ACT_) {
    constexpr int epi = decltype(EPI_)::value, act = decltype(ACT_)::value;
    if constexpr (epi == DVAE_EPI_ATOMIC) {
      // bias into the accumulators FIRST: a bias load pending beside the atomics (both count in vmcnt) made hipcc put
      // `s_waitcnt vmcnt(0)` in front of every one of the 128 atomics of a lane — each waited for the one before it
      if (add_bias) {
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) {
          const int col = n0 + wn * 64 + nt * 32 + l31;
          const float bv = col < p.N ? p.bias[col] : 0.f;
#pragma unroll
          for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] += bv;
        }
      }
    }
#pragma unroll
    for (int half = 0; half < MTW / 2; ++half) {     // 64 rows = one BatchNorm statistics chunk
      float bst[NTW][4];
      int bmod = 0;
      if constexpr (BNS) {
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
          for (int q = 0; q < 4; ++q) bst[nt][q] = 0.f;
        bmod = (m0 + wm * 128 + half * 64 + 4 * kh) % p.bn_nseg;
      }
#pragma unroll
      for (int m2 = 0; m2 < 2; ++m2) {
        const int mt = 2 * half + m2;
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) {
          const int col = n0 + wn * 64 + nt * 32 + l31;
          if (col >= p.N) continue;            // (N % 4 == 0: a quad of lanes is inside or outside as a whole)
          const float bias_v = (epi != DVAE_EPI_ATOMIC && add_bias) ? p.bias[col] : 0.f;
          const int row0 = m0 + wm * 128 + mt * 32 + 4 * kh;
          float* cbase = C + (int64_t)row0 * p.ldc + col;
          // Stores as 16 bytes per lane (a wave-instruction costs the store path ~70 cycles whatever its width).  The
          // accumulator holds 4 consecutive ROWS of one column per lane (r & 3); a 4 x 4 transpose inside each quad of
          // lanes (two DPP butterfly stages) turns them into 4 consecutive COLUMNS of one row.  Atomic accumulation
          // keeps the scalar form (one 128-byte segment per row and instruction).
          const int q4 = lane & 3;
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            float x[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const int r = 4 * g + e, dr = e + 8 * g;
              const bool ok = row0 + dr < p.M;
              float v = acc[mt][nt][r] + bias_v;
              v = act_apply(v, act);     // (act is a constant here)
              if constexpr (epi == DVAE_EPI_ATOMIC) {
                if (ok) atomicAdd(cbase + (int64_t)dr * p.ldc, v);
              }
              x[e] = v;
              if constexpr (BNS && BNU) {
                const float uu = ok ? v : 0.f;
                bst[nt][0] += uu;
                bst[nt][1] += uu * uu;
              } else if constexpr (BNS) {
                const int nseg = p.bn_nseg;
                int rm = bmod + m2 * 32 + dr;
                if (nseg >= 64) rm -= (rm >= nseg) ? nseg : 0; else rm %= nseg;
                const float uu = ok ? v : 0.f;
                const bool g1 = rm >= nseg / p.bn_groups;
                bst[nt][0] += g1 ? 0.f : uu;
                bst[nt][1] += g1 ? 0.f : uu * uu;
                bst[nt][2] += g1 ? uu : 0.f;
                bst[nt][3] += g1 ? uu * uu : 0.f;
              }
            }
            if constexpr (epi != DVAE_EPI_ATOMIC) {
              // stage 1: lane bit 0 <-> element bit 0 (quad_perm [1,0,3,2] = 0xB1); stage 2: bit 1 (quad_perm [2,3,0,1] = 0x4E)
              float y[4], z[4];
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const float o = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x[e ^ 1]), 0xB1, 0xF, 0xF, true));
                y[e] = ((q4 ^ e) & 1) ? o : x[e];
              }
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const float o = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, y[e ^ 2]), 0x4E, 0xF, 0xF, true));
                z[e] = ((q4 ^ e) & 2) ? o : y[e];
              }
              // this lane now holds row (8 g + 4 kh + q4) of the tile, columns 4 (l31 >> 2) .. + 3
              const int row = row0 + 8 * g + q4;
              if (row < p.M) {
                f32x4* dst = reinterpret_cast<f32x4*>(C + (int64_t)row * p.ldc + (col - q4));
                f32x4 o4 = {z[0], z[1], z[2], z[3]};
                if constexpr (epi == DVAE_EPI_ACCUM) o4 += *dst;
                *dst = o4;
              }
            }
          }
        }
      }
      if constexpr (BNS) {
        const int chunk = tile_m * 4 + wm * 2 + half, nchunks = (p.M + DVAE_BN_ROWS_PER_CHUNK - 1) / DVAE_BN_ROWS_PER_CHUNK;
        bool second = false;      // BNU: the chunk's sums belong to the second group
        if constexpr (BNU) second = ((m0 + wm * 128 + half * 64) % p.bn_nseg) >= p.bn_nseg / p.bn_groups;
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) {
          if (second) {
            bst[nt][2] = bst[nt][0];
            bst[nt][3] = bst[nt][1];
            bst[nt][0] = bst[nt][1] = 0.f;
          }
          const int col = n0 + wn * 64 + nt * 32 + l31;
#pragma unroll
          for (int q = 0; q < 4; ++q) bst[nt][q] += __shfl_xor(bst[nt][q], 32, 64);
          if (kh == 0 && col < p.N && chunk < nchunks) {
            double* o = p.bn_part + ((int64_t)chunk * p.bn_groups * p.N + col) * 2;
            o[0] = (double)bst[nt][0];
            o[1] = (double)bst[nt][1];
            if (p.bn_groups > 1) {
              o[(int64_t)p.N * 2] = (double)bst[nt][2];
              o[(int64_t)p.N * 2 + 1] = (double)bst[nt][3];
            }
          }
        }
      }
    }
  };
  using std::integral_constant;
  const int epi_here = to_slab ? DVAE_EPI_STORE : p.epi;
  if (epi_here == DVAE_EPI_ATOMIC) emit(integral_constant<int, DVAE_EPI_ATOMIC>{}, integral_constant<int, DVAE_ACT_NONE>{});
  else if (epi_here == DVAE_EPI_ACCUM) emit(integral_constant<int, DVAE_EPI_ACCUM>{}, integral_constant<int, DVAE_ACT_NONE>{});
  else if (p.act == DVAE_ACT_NONE) emit(integral_constant<int, DVAE_EPI_STORE>{}, integral_constant<int, DVAE_ACT_NONE>{});
  else if (p.act == DVAE_ACT_RELU) emit(integral_constant<int, DVAE_EPI_STORE>{}, integral_constant<int, DVAE_ACT_RELU>{});
  else emit(integral_constant<int, DVAE_EPI_STORE>{}, integral_constant<int, DVAE_ACT_TANH>{});
#ifdef DVAE_GEMM_TS2
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // every store / atomic of this wave acknowledged
  const unsigned flat_wg = blockIdx.z * gridDim.x + blockIdx.x;
  if ((threadIdx.x & 63) == 0 && flat_wg < 256) {
    unsigned long long* o = g_gemm_ts + (flat_wg * 4 + wave) * 8;
    o[0] = t2_entry; o[1] = t2_loop0; o[2] = t2_loop1; o[3] = __builtin_amdgcn_s_memrealtime();
    o[4] = (unsigned long long)n_iters;
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    o[5] = xcc & 0xf;
  }
#endif
}



// ------------------------------------------------------------------------------------------------------------------
// bf16 mode, both operands bf16 IN MEMORY: the tall structure (256 x 128 tile, one workgroup of four waves per CU, each
// wave 128 x 64 = 4 x 2 MFMA tiles, every MFMA followed by a pinned handful of the next tiles' staging instructions)
// with nothing to convert: staging is buffer_load_dwordx4 (8 bf16) -> ds_write_b128.  64-deep k-tiles = four 16-deep
// sub-steps of 8 MFMAs; fragments are read ONE SUB-STEP ahead (two fragment sets of 24 registers), k-tile i+1 is staged
// during sub-steps 0-2 of iteration i and the iteration's single barrier sits BEFORE sub-step 3, whose fragment reads are
// the first of tile i+1: a buffer is dead (no reader) from that barrier of iteration i-1... i.e. two LDS buffers suffice.
// Global loads run two k-tiles ahead of their staging (a piece's registers are reloaded right after its ds_write).
//   iteration i   sub-steps 0-2: MFMAs of tile i; reads of its next sub-step; ds_write of tile i+1 -> buffer (i+1)%2;
//                                buffer loads of tile i+3
//                 barrier
//                 sub-step 3:    MFMAs; fragment reads of tile i+1, sub-step 0
// Requires k ranges that are multiples of 64 and operands < 1 GiB (else the 128 x 128 kernel).
template <bool A_KC, bool B_KC, bool BNS, bool BNU = false>      // BNU: see gemm_x3_tall_kernel
__global__ __launch_bounds__(256) void gemm_bf16_tall_kernel(const GemmParams p) {
  constexpr int BM = 256, BN = 128, BK = 64, NTHR = 256, MTW = 4, NTW = 2;
  constexpr int LD_KC = BK + 8;
  constexpr int LDA = A_KC ? LD_KC : BM + 32;
  constexpr int LDB = B_KC ? LD_KC : BN + 32;
  constexpr int A_SZ = A_KC ? BM * LD_KC : BK * LDA;
  constexpr int B_SZ = B_KC ? BN * LD_KC : BK * LDB;
  constexpr int NPA = BM * BK / 8 / NTHR, NPB = BN * BK / 8 / NTHR;   // 16-byte pieces per thread per k-tile: 8, 4
  constexpr int NP = NPA + NPB;
  constexpr unsigned OOB = 0xC0000000u;
  __shared__ __attribute__((aligned(16))) __bf16 As[2][A_SZ];
  __shared__ __attribute__((aligned(16))) __bf16 Bs[2][B_SZ];
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

  const int t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, kh = lane >> 5;

  int tile_m, tile_n;
  gemm_tile_of(p, tile_m, tile_n);
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  int tap_fixed = 0, ks = blockIdx.z;
  if (p.tap_mode == 2) {
    tap_fixed = blockIdx.z % p.taps;
    ks = blockIdx.z / p.taps;
  }
  const int k_begin = ks * p.k_per_split;
  const int k_end = min(p.K, k_begin + p.k_per_split);
  const int kiters = (k_end - k_begin) / BK;
  const int n_iters = (p.tap_mode == 1 ? p.taps : 1) * kiters;
  const bool to_slab = p.slab != nullptr && p.split_k > 1;      // k-split without atomics: see gemm_f32_kernel
  float* __restrict__ C = (to_slab ? p.slab + (int64_t)ks * p.slab_stride : (float*)p.C) +
                          (p.tap_mode == 2 ? (int64_t)tap_fixed * p.c_tap_stride : 0);

  // operands as raw buffers (see gemm_x3_tall_kernel): byte offsets, anything outside reads zeros
  const int a_rows = A_KC ? p.M : p.K, b_rows = B_KC ? p.N : p.K;
  const __amdgpu_buffer_rsrc_t a_rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, (int)((int64_t)a_rows * p.lda * 2), 0x00020000);
  const int64_t b_bytes = (int64_t)b_rows * p.ldb * 2 * ((p.tap_mode == 1) ? p.taps : 1);
  const __amdgpu_buffer_rsrc_t b_rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.B, 0, (int)b_bytes, 0x00020000);
  unsigned vo[NP];     // byte offset of the piece in k-tile 0 (tap 2); pieces 0..NPA-1 = A, the rest B
  int off[NP];         // LDS element offset of the piece
#pragma unroll
  for (int j = 0; j < NPA; ++j) {
    const int idx = t + NTHR * j;
    if (A_KC) {
      const int row = idx >> 3, k8 = (idx & 7) * 8;
      off[j] = row * LDA + k8;
      vo[j] = (unsigned)(((int64_t)(m0 + row) * p.lda + k_begin + k8) * 2);
      if (m0 + row >= p.M) vo[j] = OOB;
    } else {
      const int kr = idx / (BM / 8), m8 = (idx % (BM / 8)) * 8;
      off[j] = kr * LDA + m8;
      vo[j] = (unsigned)(((int64_t)(k_begin + kr) * p.lda + m0 + m8) * 2);
      if (m0 + m8 >= p.M) vo[j] = OOB;
    }
  }
  const int64_t b_shift = (p.tap_mode == 2) ? (int64_t)(tap_fixed - 2) * p.bk_row_shift : 0;
#pragma unroll
  for (int j = 0; j < NPB; ++j) {
    const int idx = t + NTHR * j;
    if (B_KC) {
      const int row = idx >> 3, k8 = (idx & 7) * 8;
      off[NPA + j] = row * LDB + k8;
      vo[NPA + j] = (unsigned)(((int64_t)(n0 + row) * p.ldb + k_begin + k8) * 2);
      if (n0 + row >= p.N) vo[NPA + j] = OOB;
    } else {
      const int kr = idx / (BN / 8), n8 = (idx % (BN / 8)) * 8;
      off[NPA + j] = kr * LDB + n8;
      vo[NPA + j] = (unsigned)(((int64_t)(k_begin + kr) + b_shift) * p.ldb * 2 + (int64_t)(n0 + n8) * 2);
      if (n0 + n8 >= p.N) vo[NPA + j] = OOB;
    }
  }
  const int ntaps_loop = (p.tap_mode == 1) ? p.taps : 1;
  const int a_tap_step = (p.tap_mode == 1) ? (int)(p.a_row_shift * p.lda * 2) : 0;
  const int b_tap_step = (p.tap_mode == 1) ? (int)(p.b_tap_stride * 2) : 0;
  const int a_k_step = A_KC ? BK * 2 : (int)(BK * p.lda * 2);
  const int b_k_step = B_KC ? BK * 2 : (int)(BK * p.ldb * 2);
  int tap_n = 0, kit_n = 0;       // next tile to fetch
  unsigned a_s = 0, b_s = 0;      // its scalar byte offsets (uniform)
  auto next_tile_offsets = [&]() {
    const bool live = kit_n < kiters;
    a_s = live ? (unsigned)((tap_n - 2) * a_tap_step + kit_n * a_k_step) : OOB;
    b_s = live ? (unsigned)(tap_n * b_tap_step + kit_n * b_k_step) : OOB;
    const bool wrap = (tap_n + 1 == ntaps_loop);
    tap_n = wrap ? 0 : tap_n + 1;
    kit_n += wrap ? 1 : 0;
  };
  u32x4 gr[2][NP];
  auto load_piece = [&](u32x4& dst, int j) {
    if (j < NPA) dst = __builtin_amdgcn_raw_buffer_load_b128(a_rs, vo[j] + a_s, 0, 0);
    else dst = __builtin_amdgcn_raw_buffer_load_b128(b_rs, vo[j] + b_s, 0, 0);
  };
  auto stage_piece = [&](int buf, const u32x4& v, int j) {
    __bf16* img = j < NPA ? &As[buf][off[j]] : &Bs[buf][off[j]];
    *reinterpret_cast<u32x4*>(img) = v;
  };

  const int g16 = lane >> 4, li = lane & 15;
  const int tr_k = 8 * (g16 >> 1) + (li >> 2), tr_r = 16 * (g16 & 1) + 4 * (li & 3);
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  auto frag = [&](const __bf16* img, bool kc, int ld, int row0, int sub) -> bf16x8 {
    if (kc) return *reinterpret_cast<const bf16x8*>(&img[(row0 + l31) * ld + 16 * sub + 8 * kh]);
    const __bf16* q0 = &img[(16 * sub + tr_k) * ld + row0 + tr_r];
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(q0));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(q0 + 4 * ld));
    return __builtin_shufflevector(__builtin_bit_cast(bf16x4, lo), __builtin_bit_cast(bf16x4, hi), 0, 1, 2, 3, 4, 5, 6, 7);
  };
  struct Frags {
    bf16x8 a[MTW], b[NTW];
  };
  auto read_frag = [&](Frags& f, int buf, int sub, int w) {    // w = 0..5: A tiles, then B tiles
    if (w < MTW) f.a[w] = frag(&As[buf][0], A_KC, LDA, wm * 128 + w * 32, sub);
    else f.b[w - MTW] = frag(&Bs[buf][0], B_KC, LDB, wn * 64 + (w - MTW) * 32, sub);
  };

  f32x16 acc[MTW][NTW];
#pragma unroll
  for (int i = 0; i < MTW; ++i)
#pragma unroll
    for (int j = 0; j < NTW; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  Frags fr[2];
  // ---- prologue: tile 0 staged, tiles 1 and 2 in flight, fragments of (tile 0, sub-step 0) read
  next_tile_offsets();
#pragma unroll
  for (int j = 0; j < NP; ++j) load_piece(gr[0][j], j);
#pragma unroll
  for (int j = 0; j < NP; ++j) stage_piece(0, gr[0][j], j);
  next_tile_offsets();
#pragma unroll
  for (int j = 0; j < NP; ++j) load_piece(gr[1][j], j);     // tile 1 -> set 1
  next_tile_offsets();
#pragma unroll
  for (int j = 0; j < NP; ++j) load_piece(gr[0][j], j);     // tile 2 -> set 0
  __syncthreads();
#pragma unroll
  for (int w = 0; w < MTW + NTW; ++w) read_frag(fr[0], 0, 0, w);

  // one k-tile = 32 pinned steps (see gemm_x3_tall_kernel): step g = MFMA g, then at most three staging instructions
  auto step = [&](auto G, auto U) {
    constexpr int g = decltype(G)::value, u = decltype(U)::value;    // u = parity of the k-tile being computed
    constexpr int sub = g / 8, e = g % 8, mt = e / 2, nt = e % 2;
    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[sub & 1].a[mt], fr[sub & 1].b[nt], acc[mt][nt], 0, 0, 0);
    // fragments of the next sub-step (the first sub-step of tile i+1 after the barrier)
    if constexpr (e < MTW + NTW) {
      if constexpr (sub < 3) read_frag(fr[(sub + 1) & 1], u, sub + 1, e);
      else read_frag(fr[0], u ^ 1, 0, e);
    }
    // staging of tile i+1 (set u^1 -> buffer u^1) and reload of the set with tile i+3: four pieces per sub-step 0-2
    if constexpr (sub < 3) {
      constexpr int j = 4 * sub + e / 2;
      if constexpr (e % 2 == 0) stage_piece(u ^ 1, gr[u ^ 1][j], j);
      else load_piece(gr[u ^ 1][j], j);
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  auto run = [&](auto U) {
    for_seq([&](auto G) { step(G, U); }, std::make_integer_sequence<int, 24>{});
    __syncthreads();
    __builtin_amdgcn_sched_barrier(0);
    for_seq([&](auto G) { step(std::integral_constant<int, 24 + decltype(G)::value>{}, U); }, std::make_integer_sequence<int, 8>{});
  };
  for (int it0 = 0; it0 < n_iters; it0 += 2) {
    next_tile_offsets();      // tile it0 + 3
    run(std::integral_constant<int, 0>{});
    next_tile_offsets();
    run(std::integral_constant<int, 1>{});
  }

  // ---- epilogue: C/D lane map of the 32x32 tile: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5).
  // One straight-line copy per (epilogue kind, activation): with `epi` / `act` tested per element the 128 values of a
  // lane went through ~1 000 scalar branches — 37 k cycles per tile (in-kernel stamps), a quarter of a 64-step k loop.
  const bool add_bias = (p.bias != nullptr) && (ks == 0);
  auto emit = [&](auto EPI_, auto ACT_) {
    constexpr int epi = decltype(EPI_)::value, act = decltype(ACT_)::value;
    if constexpr (epi == DVAE_EPI_ATOMIC) {
      // bias into the accumulators FIRST: a bias load pending beside the atomics (both count in vmcnt) made hipcc put
      // `s_waitcnt vmcnt(0)` in front of every one of the 128 atomics of a lane — each waited for the one before it
      if (add_bias) {
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) {
          const int col = n0 + wn * 64 + nt * 32 + l31;
          const float bv = col < p.N ? p.bias[col] : 0.f;
#pragma unroll
          for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] += bv;
        }
      }
    }
#pragma unroll
    for (int half = 0; half < MTW / 2; ++half) {     // 64 rows = one BatchNorm statistics chunk
      float bst[NTW][4];
      int bmod = 0;
      if constexpr (BNS) {
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
          for (int q = 0; q < 4; ++q) bst[nt][q] = 0.f;
        bmod = (m0 + wm * 128 + half * 64 + 4 * kh) % p.bn_nseg;
      }
#pragma unroll
      for (int m2 = 0; m2 < 2; ++m2) {
        const int mt = 2 * half + m2;
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) {
          const int col = n0 + wn * 64 + nt * 32 + l31;
          if (col >= p.N) continue;            // (N % 4 == 0: a quad of lanes is inside or outside as a whole)
          const float bias_v = (epi != DVAE_EPI_ATOMIC && add_bias) ? p.bias[col] : 0.f;
          const int row0 = m0 + wm * 128 + mt * 32 + 4 * kh;
          float* cbase = C + (int64_t)row0 * p.ldc + col;
          // Stores as 16 bytes per lane (a wave-instruction costs the store path ~70 cycles whatever its width).  The
          // accumulator holds 4 consecutive ROWS of one column per lane (r & 3); a 4 x 4 transpose inside each quad of
          // lanes (two DPP butterfly stages) turns them into 4 consecutive COLUMNS of one row.  Atomic accumulation
          // keeps the scalar form (one 128-byte segment per row and instruction).
          const int q4 = lane & 3;
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            float x[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const int r = 4 * g + e, dr = e + 8 * g;
              const bool ok = row0 + dr < p.M;
              float v = acc[mt][nt][r] + bias_v;
              v = act_apply(v, act);     // (act is a constant here)
              if constexpr (epi == DVAE_EPI_ATOMIC) {
                if (ok) atomicAdd(cbase + (int64_t)dr * p.ldc, v);
              }
              x[e] = v;
              if constexpr (BNS && BNU) {
                const float uu = ok ? v : 0.f;
                bst[nt][0] += uu;
                bst[nt][1] += uu * uu;
              } else if constexpr (BNS) {
                const int nseg = p.bn_nseg;
                int rm = bmod + m2 * 32 + dr;
                if (nseg >= 64) rm -= (rm >= nseg) ? nseg : 0; else rm %= nseg;
                const float uu = ok ? v : 0.f;
                const bool g1 = rm >= nseg / p.bn_groups;
                bst[nt][0] += g1 ? 0.f : uu;
                bst[nt][1] += g1 ? 0.f : uu * uu;
                bst[nt][2] += g1 ? uu : 0.f;
                bst[nt][3] += g1 ? uu * uu : 0.f;
              }
            }
            if constexpr (epi != DVAE_EPI_ATOMIC) {
              // stage 1: lane bit 0 <-> element bit 0 (quad_perm [1,0,3,2] = 0xB1); stage 2: bit 1 (quad_perm [2,3,0,1] = 0x4E)
              float y[4], z[4];
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const float o = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x[e ^ 1]), 0xB1, 0xF, 0xF, true));
                y[e] = ((q4 ^ e) & 1) ? o : x[e];
              }
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const float o = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, y[e ^ 2]), 0x4E, 0xF, 0xF, true));
                z[e] = ((q4 ^ e) & 2) ? o : y[e];
              }
              // this lane now holds row (8 g + 4 kh + q4) of the tile, columns 4 (l31 >> 2) .. + 3
              const int row = row0 + 8 * g + q4;
              if (row < p.M) {
                f32x4* dst = reinterpret_cast<f32x4*>(C + (int64_t)row * p.ldc + (col - q4));
                f32x4 o4 = {z[0], z[1], z[2], z[3]};
                if constexpr (epi == DVAE_EPI_ACCUM) o4 += *dst;
                *dst = o4;
              }
            }
          }
        }
      }
      if constexpr (BNS) {
        const int chunk = tile_m * 4 + wm * 2 + half, nchunks = (p.M + DVAE_BN_ROWS_PER_CHUNK - 1) / DVAE_BN_ROWS_PER_CHUNK;
        bool second = false;      // BNU: the chunk's sums belong to the second group
        if constexpr (BNU) second = ((m0 + wm * 128 + half * 64) % p.bn_nseg) >= p.bn_nseg / p.bn_groups;
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) {
          if (second) {
            bst[nt][2] = bst[nt][0];
            bst[nt][3] = bst[nt][1];
            bst[nt][0] = bst[nt][1] = 0.f;
          }
          const int col = n0 + wn * 64 + nt * 32 + l31;
#pragma unroll
          for (int q = 0; q < 4; ++q) bst[nt][q] += __shfl_xor(bst[nt][q], 32, 64);
          if (kh == 0 && col < p.N && chunk < nchunks) {
            double* o = p.bn_part + ((int64_t)chunk * p.bn_groups * p.N + col) * 2;
            o[0] = (double)bst[nt][0];
            o[1] = (double)bst[nt][1];
            if (p.bn_groups > 1) {
              o[(int64_t)p.N * 2] = (double)bst[nt][2];
              o[(int64_t)p.N * 2 + 1] = (double)bst[nt][3];
            }
          }
        }
      }
    }
  };
  using std::integral_constant;
  const int epi_here = to_slab ? DVAE_EPI_STORE : p.epi;
  if (epi_here == DVAE_EPI_ATOMIC) emit(integral_constant<int, DVAE_EPI_ATOMIC>{}, integral_constant<int, DVAE_ACT_NONE>{});
  else if (epi_here == DVAE_EPI_ACCUM) emit(integral_constant<int, DVAE_EPI_ACCUM>{}, integral_constant<int, DVAE_ACT_NONE>{});
  else if (p.act == DVAE_ACT_NONE) emit(integral_constant<int, DVAE_EPI_STORE>{}, integral_constant<int, DVAE_ACT_NONE>{});
  else if (p.act == DVAE_ACT_RELU) emit(integral_constant<int, DVAE_EPI_STORE>{}, integral_constant<int, DVAE_ACT_RELU>{});
  else emit(integral_constant<int, DVAE_EPI_STORE>{}, integral_constant<int, DVAE_ACT_TANH>{});
}

// conv forward with BatchNorm statistics (k-contiguous operands, 128-row tiles, unsplit, plain store)
void launch_bns(const GemmParams& p, dim3 grid, hipStream_t s, bool narrow, int bk, int mode) {
#define BNS_LAUNCH(NTW_, BK_, MODE_) hipLaunchKernelGGL((gemm_f32_kernel<true, true, NTW_, BK_, 2, MODE_, true>), grid, dim3(256), 0, s, p)
  if (mode == DVAE_MODE_BF16) {
#define BNS16(NTW_, A16_, B16_) hipLaunchKernelGGL((gemm_f32_kernel<true, true, NTW_, 32, 2, 1, true, A16_, B16_>), grid, dim3(256), 0, s, p)
#define BNS16_N(A16_, B16_) do { if (narrow) BNS16(1, A16_, B16_); else BNS16(2, A16_, B16_); } while (0)
    if (p.a16 && p.b16) BNS16_N(true, true); else if (p.a16) BNS16_N(true, false); else if (p.b16) BNS16_N(false, true);
    else BNS16_N(false, false);
#undef BNS16_N
#undef BNS16
  }
  else if (mode == DVAE_MODE_F32X3) { if (narrow) BNS_LAUNCH(1, 16, 2); else BNS_LAUNCH(2, 16, 2); }
  else if (bk == 32) { if (narrow) BNS_LAUNCH(1, 32, 0); else BNS_LAUNCH(2, 32, 0); }
  else { if (narrow) BNS_LAUNCH(1, 16, 0); else BNS_LAUNCH(2, 16, 0); }
#undef BNS_LAUNCH
}

template <bool AK, bool BKC>
void launch_variant(const GemmParams& p, dim3 grid, hipStream_t s, bool narrow, int bk, bool big, int mode) {
  if (mode == DVAE_MODE_BF16) {   // bf16 operands (rounded while staged, or already bf16 in memory), fp32 accumulation
#define BF16K(NTW_, A16_, B16_) hipLaunchKernelGGL((gemm_f32_kernel<AK, BKC, NTW_, 32, 2, 1, false, A16_, B16_>), grid, dim3(256), 0, s, p)
#define BF16K_N(A16_, B16_) do { if (narrow) BF16K(1, A16_, B16_); else BF16K(2, A16_, B16_); } while (0)
    if (p.a16 && p.b16) BF16K_N(true, true); else if (p.a16) BF16K_N(true, false); else if (p.b16) BF16K_N(false, true);
    else BF16K_N(false, false);
#undef BF16K_N
#undef BF16K
    return;
  }
  if (mode == DVAE_MODE_F32X3) {  // fp32 operands split into 3 bf16 terms, 6 bf16 MFMAs per product
    if (narrow) hipLaunchKernelGGL((gemm_f32_kernel<AK, BKC, 1, 16, 2, 2>), grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL((gemm_f32_kernel<AK, BKC, 2, 16, 2, 2>), grid, dim3(256), 0, s, p);
    return;
  }
  if (big) {   // 256 x 256 x 32 tile, 16 waves
    hipLaunchKernelGGL((gemm_f32_kernel<AK, BKC, 2, 32, 4>), grid, dim3(1024), 0, s, p);
    return;
  }
  dim3 block(256);
  if (bk == 32) {
    if (narrow) hipLaunchKernelGGL((gemm_f32_kernel<AK, BKC, 1, 32, 2>), grid, block, 0, s, p);
    else hipLaunchKernelGGL((gemm_f32_kernel<AK, BKC, 2, 32, 2>), grid, block, 0, s, p);
  } else {
    if (narrow) hipLaunchKernelGGL((gemm_f32_kernel<AK, BKC, 1, 16, 2>), grid, block, 0, s, p);
    else hipLaunchKernelGGL((gemm_f32_kernel<AK, BKC, 2, 16, 2>), grid, block, 0, s, p);
  }
}

}  // namespace
// gemm256.hip: the 256 x 256 LDS-DMA kernel of the bf16 mode (`params` is a complete GemmParams)
int dvae_launch_gemm_bf16_256(const void* params, int a_kc, int b_kc, int bn_uniform, unsigned gx, unsigned gz, hipStream_t s);
namespace {

int launch_gemm(GemmParams& p, bool a_kc, bool b_kc, int mode, hipStream_t s) {
  // operand storage flags ride in the upper bits of `mode`
  p.a16 = (mode >= 0 && (mode & DVAE_MODE_A_BF16)) ? 1 : 0;
  p.b16 = (mode >= 0 && (mode & DVAE_MODE_B_BF16)) ? 1 : 0;
  p.c16 = (mode >= 0 && (mode & DVAE_MODE_C_BF16)) ? 1 : 0;
  if (mode >= 0) mode &= 0xff;
  if (mode == DVAE_MODE_DEFAULT) mode = g_dvae_compute_mode;
  if ((p.a16 || p.b16 || p.c16) && mode != DVAE_MODE_BF16) return DVAE_EINVAL;
  if (p.c16 && (p.epi != DVAE_EPI_STORE || p.split_k > 1)) return DVAE_EINVAL;
  // a bf16 operand moves 8 elements per 16-byte load
  if (p.a16 && ((p.lda & 7) || (a_kc ? (p.K & 7) : (p.M & 7)))) return DVAE_EINVAL;
  if (p.b16 && ((p.ldb & 7) || (b_kc ? (p.K & 7) : (p.N & 7)))) return DVAE_EINVAL;
  if (p.b16 && p.tap_mode == 1 && (p.b_tap_stride & 7)) return DVAE_EINVAL;
  if (mode != DVAE_MODE_F32 && mode != DVAE_MODE_BF16 && mode != DVAE_MODE_F32X3) return DVAE_EINVAL;
  if (p.M <= 0 || p.N <= 0 || p.K <= 0) return DVAE_EINVAL;
  if (!p.A || !p.B || !p.C) return DVAE_EINVAL;
  if ((p.lda & 3) || (p.ldb & 3)) return DVAE_EINVAL;
  if ((a_kc || b_kc) && (p.K & 3)) return DVAE_EINVAL;
  if ((((uintptr_t)p.A) | ((uintptr_t)p.B)) & 15) return DVAE_EINVAL;
  if (!a_kc && (p.M & 3)) return DVAE_EINVAL;
  if (!b_kc && (p.N & 3)) return DVAE_EINVAL;
  // deterministic test mode: one writer per element, no atomic races (slab splits have one writer per element anyway)
  if (p.split_k < 1 || (g_dvae_deterministic && !p.slab)) p.split_k = 1;
  if (p.slab) {      // k-split without atomics: plain stores into caller-provided slabs (dvae_gemm_f32_slabs ...)
    if (p.slab_cap < 0 || (p.slab_stride & 3) || (((uintptr_t)p.slab) & 15) || p.c16 || p.act != DVAE_ACT_NONE ||
        (p.epi != DVAE_EPI_STORE && p.epi != DVAE_EPI_ACCUM))
      return DVAE_EINVAL;
    if (p.split_k > p.slab_cap) p.split_k = p.slab_cap > 0 ? p.slab_cap : 1;
  } else if (p.split_k > 1 && (p.epi != DVAE_EPI_ATOMIC || p.act != DVAE_ACT_NONE)) return DVAE_EINVAL;
  const bool splittable = p.slab != nullptr || p.epi == DVAE_EPI_ATOMIC;      // a launch may cut k further itself
  const int max_sk = p.slab ? (p.slab_cap > 0 ? p.slab_cap : 1) : (1 << 30);
  if (p.epi != DVAE_EPI_STORE && p.act != DVAE_ACT_NONE) return DVAE_EINVAL;
  // tuning knobs for experiments (scripts/one_shape.py): environment variables in the DEV build, constants in the product
  static const int bk_env = dvae_dev_knob("DVAE_GEMM_BK", 0);
  static const int narrow_env = dvae_dev_knob("DVAE_GEMM_NARROW", -1);
  int kps = (p.K + p.split_k - 1) / p.split_k;
  // k-tile 32 when the per-split K allows it without padding waste (a long split is simply rounded up to whole
  // 32-deep tiles: the last split takes what is left)
  int bk = ((kps % 32 == 0 && kps >= 64) || (p.split_k > 1 && kps >= 512)) ? 32 : 16;
  if (bk_env == 16 || bk_env == 32) bk = bk_env;
  const bool bf = (mode == DVAE_MODE_BF16);
  if (bf) bk = 32;                       // the bf16 kernel has one k-tile; ragged tails are zero-filled
  if (mode == DVAE_MODE_F32X3) bk = 16;  // three images per operand: 16-deep tiles keep two workgroups per CU
  kps = ((kps + bk - 1) / bk) * bk;
  p.k_per_split = kps;
  p.split_k = (p.K + kps - 1) / kps;
  if (p.batch > 1 && (p.tap_mode != 0 || p.bias || p.bn_part || p.batch > 4 || p.c16)) return DVAE_EINVAL;
  int zdim = p.split_k * (p.tap_mode == 2 ? p.taps : 1);
  // split mode: the 256 x 128 tile (gemm_x3_tall_kernel, one workgroup per CU) when it still fills the chip; an
  // atomically accumulated product (weight gradients) is cut into twice the k-splits for it
  static const int tall_env = dvae_dev_knob("DVAE_GEMM_TALL", -1);
  bool tall = false;
  // (its raw-buffer addressing: every k range a multiple of 16, operands below 1 GiB)
  const int64_t a_bytes = (int64_t)(a_kc ? p.M : p.K) * p.lda * 4;
  const int64_t b_bytes = (int64_t)(b_kc ? p.N : p.K) * p.ldb * 4 * (p.tap_mode == 1 ? p.taps : 1);
  const bool tall_ok = (p.K % 16 == 0) && a_bytes < (1ll << 30) && b_bytes < (1ll << 30) &&   // + 16-byte stores of C
                       (p.N % 4 == 0) && (p.ldc % 4 == 0) && (((uintptr_t)p.C & 15) == 0) && (p.c_tap_stride % 4 == 0);
  if (mode == DVAE_MODE_F32X3 && tall_env != 0 && tall_ok && p.M >= 256 && p.N > 64 && p.batch <= 1) {
    const int t2 = ((p.M + 255) / 256) * ((p.N + 127) / 128);
    if (t2 * zdim < 192 && p.split_k > 1 && splittable && kps >= 1024 &&
        (p.K + ((kps / 2 + bk - 1) / bk) * bk - 1) / (((kps / 2 + bk - 1) / bk) * bk) <= max_sk) {
      kps = ((kps / 2 + bk - 1) / bk) * bk;
      p.k_per_split = kps;
      p.split_k = (p.K + kps - 1) / kps;
      zdim = p.split_k * (p.tap_mode == 2 ? p.taps : 1);
    }
    // one workgroup per CU: nothing hides a tile's prologue (cold loads) and epilogue (128 KB of C per CU), so the tile
    // should be long.  Round 3 measured 24 and 12 k-steps equal and 60 slower by 0.1 ms per step; with the 16-byte store
    // epilogues of round 4 the short ones gain a little too (scripts/small_shapes.py, 17 short shapes of the step: 873 us
    // at 24, 846 at 8, 843 at 4): from 8 k-steps on
    static const int tall_min = dvae_dev_knob("DVAE_GEMM_TALL_MIN", 8);
    const int steps_per_tile = (kps / 16) * (p.tap_mode == 1 ? p.taps : 1);
    tall = (t2 * zdim >= 192 && steps_per_tile >= tall_min) || tall_env == 1;
  }
  // bf16 mode, both operands bf16 in memory, fp32 result: gemm_bf16_tall_kernel (64-deep k-tiles) under the same rules
  bool tall16 = false;
  if (bf && p.batch <= 1 && p.a16 && p.b16 && !p.c16 && tall_env != 0 && p.M >= 256 && p.N > 64 && (p.K % 64 == 0) &&
      (p.N % 4 == 0) && (p.ldc % 4 == 0) && (((uintptr_t)p.C & 15) == 0) && (p.c_tap_stride % 4 == 0) &&
      a_bytes < (1ll << 31) && b_bytes < (1ll << 31)) {      // (a_bytes / b_bytes above count 4 bytes per element)
    int kps64 = ((kps + 63) / 64) * 64;
    int sk64 = (p.K + kps64 - 1) / kps64;
    const int t2 = ((p.M + 255) / 256) * ((p.N + 127) / 128);
    if (t2 * sk64 * (p.tap_mode == 2 ? p.taps : 1) < 192 && p.split_k > 1 && splittable && kps64 >= 1024 &&
        (p.K + ((kps64 / 2 + 63) / 64) * 64 - 1) / (((kps64 / 2 + 63) / 64) * 64) <= max_sk) {
      kps64 = ((kps64 / 2 + 63) / 64) * 64;
      sk64 = (p.K + kps64 - 1) / kps64;
    }
    const int z64 = sk64 * (p.tap_mode == 2 ? p.taps : 1);
    static const int tall16_min = dvae_dev_knob("DVAE_GEMM_TALL16_MIN", 8);
    const int iters = (kps64 / 64) * (p.tap_mode == 1 ? p.taps : 1);
    if (t2 * z64 >= 192 && iters >= tall16_min) {
      tall16 = true;
      kps = kps64;
      p.k_per_split = kps;
      p.split_k = sk64;
      zdim = z64;
    }
  }
  // ... and 256 x 256 tiles staged by LDS-DMA (gemm256.hip) when THOSE still fill the chip (one workgroup per CU): both
  // operands of one layout.  An atomically accumulated product is cut into as many k-splits as make one workgroup per CU.
  bool t256 = false;
  const int t256_env = dvae_dev_knob("DVAE_GEMM_256", -1);      // (dev build: read per call, scripts/g256_check.py toggles it)
  // (the row-contiguous form — weight gradients, k-splits into slabs — from 20 output tile-taps on: 1.13 x / 1.05 x the tall
  // kernel at 64 / 32 tiles, 1.06 x on the conv weight gradients (4 tiles x 5 taps), 1.01 x at 16 tiles.  DVAE_GEMM_256=2 in
  // the dev build forces it)
  if (bf && p.batch <= 1 && p.a16 && p.b16 && !p.c16 && t256_env != 0 && a_kc == b_kc && p.M >= 256 && p.N >= 256 &&
      (p.K % 64 == 0) && (p.N % 8 == 0) && (p.M % 8 == 0) && (p.ldc % 4 == 0) && (((uintptr_t)p.C & 15) == 0) &&
      (p.c_tap_stride % 4 == 0) && a_bytes < (1ll << 31) && b_bytes < (1ll << 31)) {
    const int tz = ((p.M + 255) / 256) * ((p.N + 255) / 256) * (p.tap_mode == 2 ? p.taps : 1);
    int sk = 1;
    if (p.split_k > 1 && p.slab) sk = 256 / tz > 1 ? 256 / tz : 1;
    if (sk > max_sk) sk = max_sk;
    int kps256 = (((p.K + sk - 1) / sk + 63) / 64) * 64;
    if (kps256 < 512 && sk > 1) kps256 = 512;
    sk = (p.K + kps256 - 1) / kps256;
    const int iters = (kps256 / 64) * (p.tap_mode == 1 ? p.taps : 1);
    const int t256_min = dvae_dev_knob("DVAE_GEMM_256_MIN", 224);
    // (no atomic and no tanh epilogue in that kernel: k-splits only into slabs)
    const bool epi_ok = p.epi != DVAE_EPI_ATOMIC && p.act != DVAE_ACT_TANH && (p.split_k == 1 || p.slab != nullptr) &&
                        (a_kc || tz >= 20 || t256_env == 2);
    if (epi_ok && ((tz * sk >= t256_min && iters >= 8) || t256_env >= 1)) {
      t256 = true;
      tall16 = false;
      kps = kps256;
      p.k_per_split = kps;
      p.split_k = sk;
      zdim = sk * (p.tap_mode == 2 ? p.taps : 1);
    }
  }
  static const int big_env = dvae_dev_knob("DVAE_GEMM_BIG", -1);
  // 256x256 tiles when they still give every CU >= 2 workgroups' worth of tiles and the k-tile can be 32
  const int tiles256 = ((p.M + 255) / 256) * ((p.N + 255) / 256) * zdim;
  bool big = (bk == 32) && (tiles256 >= 512) && (p.N >= 256) && (p.M >= 256) && (p.tap_mode == 0) && (p.batch <= 1);
  if (big_env >= 0) big = (big_env != 0) && (bk == 32) && (p.tap_mode == 0) && (p.batch <= 1);
  if (mode != DVAE_MODE_F32) big = false;   // the 16-wave tile exists for the fp32 MFMA only (128 registers per lane)
  const int bm = (big || tall || tall16 || t256) ? 256 : 128;
  p.tiles_m = (p.M + bm - 1) / bm;
  // 128-wide n-tiles unless that leaves the chip badly under-filled: then 64-wide
  const int tiles128 = p.tiles_m * ((p.N + 127) / 128) * zdim;
  const bool narrow = !big && !tall && !tall16 && !t256 && (narrow_env >= 0 ? (narrow_env != 0) : (tiles128 < 256 && p.N > 32));
  const int bn = (big || t256) ? 256 : (narrow ? 64 : 128);
  const int tiles_n = (p.N + bn - 1) / bn;
  static const int xcd_env = dvae_dev_knob("DVAE_GEMM_XCDMAP", 1);
  static const int strip_env = dvae_dev_knob("DVAE_GEMM_STRIP", 8);    // n-tiles per block (0: one, the round-2 map)
  p.xcd_map = (xcd_env && (p.tiles_m % 8 == 0) && !big && (xcd_env == 2 || p.tap_mode == 1 || strip_env > 0)) ? 1 : 0;
  {
    // the block resident on one XCD: 32 CUs x (1 tall | 2 square) workgroups; fabric bytes per block ~ gm * BM + gn * BN
    const int per = p.tiles_m >> 3, conc = (tall || tall16 || t256) ? 32 : 64;
    int gn = 1;
    for (int d = 1; d <= (strip_env > 0 ? strip_env : 1); ++d)
      if (tiles_n % d == 0) gn = d;
    int gm = 1;
    for (int d = 1; d <= conc / gn && d <= per; ++d)
      if (per % d == 0) gm = d;
    if (strip_env <= 0) gm = per > 0 ? per : 1;
    p.map_gm = gm;
    p.map_gn = gn;
    p.map_nstr = tiles_n / gn;
  }
  const int nb = p.batch > 1 ? p.batch : 1;
  p.c_vec = (!p.c16 && (p.N % 4 == 0) && (p.ldc % 4 == 0) && (((uintptr_t)p.C & 15) == 0) && (p.c_tap_stride % 4 == 0)) ? 1 : 0;
  if (p.batch > 1)
    for (int b = 0; b < p.batch; ++b) p.c_vec &= ((p.c_boff[b] & 15) == 0) ? 1 : 0;
  dim3 grid(p.tiles_m * tiles_n, 1, zdim * nb);
  // tag of this instantiation: template arguments <A_KC, B_KC, NTW, BK, WG, MODE> + tap mode (dvae_prof_collect_tags)
  const unsigned tag = (a_kc ? 1u : 0u) | (b_kc ? 2u : 0u) | ((narrow ? 1u : 2u) << 2) | ((unsigned)bk << 4) |
                       ((big ? 4u : t256 ? 3u : (tall || tall16) ? 1u : 2u) << 10) | ((unsigned)mode << 13) | ((unsigned)p.tap_mode << 15) |
                       ((unsigned)p.a16 << 17) | ((unsigned)p.b16 << 18) | ((p.bn_part ? 1u : 0u) << 19);
  // algorithmic bytes: every operand element once (the activation matrix of a conv once, not once per tap)
  const double ntap = p.tap_mode ? p.taps : 1;
  const double ea = p.a16 ? 2.0 : 4.0, eb = p.b16 ? 2.0 : 4.0;   // bf16 mode: operands that are bf16 in memory
  const double alg_bytes = ea * (double)p.M * p.K + eb * (double)p.K * p.N * (p.tap_mode == 1 ? ntap : 1.0) +
                           (p.c16 ? 2.0 : 4.0) * (double)p.M * p.N * (p.tap_mode == 2 ? ntap : 1.0);
  ProfScope prof(1, s, 2.0 * p.M * p.N * (double)p.K * ntap * nb, tag, alg_bytes * nb);
  if (p.bn_part && (!a_kc || !b_kc || big || p.split_k != 1 || p.epi != DVAE_EPI_STORE || p.act != DVAE_ACT_NONE))
    return DVAE_EINVAL;
  // statistics chunks (64 rows) that never straddle a group or a frame: the cheap form of the BatchNorm epilogue
  const bool bn_uniform = p.bn_part && p.bn_groups >= 1 && ((p.bn_nseg / p.bn_groups) % 64 == 0) && (p.bn_nseg % p.bn_groups == 0);
  if (t256) {
    const int rc = dvae_launch_gemm_bf16_256(&p, a_kc, b_kc, bn_uniform, grid.x, grid.z, s);
    if (rc != DVAE_OK) return rc;
  } else if (tall16) {
#define TALL16(AK_, BK_, BNS_) hipLaunchKernelGGL((gemm_bf16_tall_kernel<AK_, BK_, BNS_>), grid, dim3(256), 0, s, p)
    if (p.bn_part && bn_uniform) hipLaunchKernelGGL((gemm_bf16_tall_kernel<true, true, true, true>), grid, dim3(256), 0, s, p);
    else if (p.bn_part) TALL16(true, true, true);
    else if (a_kc && b_kc) TALL16(true, true, false);
    else if (a_kc && !b_kc) TALL16(true, false, false);
    else if (!a_kc && b_kc) TALL16(false, true, false);
    else TALL16(false, false, false);
#undef TALL16
  } else if (tall) {
#define TALL(AK_, BK_, BNS_) hipLaunchKernelGGL((gemm_x3_tall_kernel<AK_, BK_, BNS_>), grid, dim3(256), 0, s, p)
    if (p.bn_part && bn_uniform) hipLaunchKernelGGL((gemm_x3_tall_kernel<true, true, true, true>), grid, dim3(256), 0, s, p);
    else if (p.bn_part) TALL(true, true, true);
    else if (a_kc && b_kc) TALL(true, true, false);
    else if (a_kc && !b_kc) TALL(true, false, false);
    else if (!a_kc && b_kc) TALL(false, true, false);
    else TALL(false, false, false);
#undef TALL
  } else if (p.bn_part) {
    launch_bns(p, grid, s, narrow, bk, mode);
  } else if (a_kc && b_kc)
    launch_variant<true, true>(p, grid, s, narrow, bk, big, mode);
  else if (a_kc && !b_kc)
    launch_variant<true, false>(p, grid, s, narrow, bk, big, mode);
  else if (!a_kc && b_kc)
    launch_variant<false, true>(p, grid, s, narrow, bk, big, mode);
  else
    launch_variant<false, false>(p, grid, s, narrow, bk, big, mode);
  return dvae_check_launch();
}

}  // namespace

#if defined(DVAE_GEMM_TS) || defined(DVAE_GEMM_TS2)
DVAE_API int dvae_probe_gemm_timeline(unsigned long long* host_out, int n_words) {
  if (hipDeviceSynchronize() != hipSuccess) return DVAE_ELAUNCH;
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_gemm_ts), sizeof(unsigned long long) * n_words) == hipSuccess ? 0 : DVAE_ELAUNCH;
}
#endif

DVAE_API int dvae_set_compute_mode(int mode) {
  if (mode != DVAE_MODE_F32 && mode != DVAE_MODE_BF16 && mode != DVAE_MODE_F32X3) return DVAE_EINVAL;
  g_dvae_compute_mode = mode;
  return DVAE_OK;
}
DVAE_API int dvae_get_compute_mode(void) { return g_dvae_compute_mode; }
DVAE_API int dvae_set_deterministic(int on) {
  g_dvae_deterministic = on ? 1 : 0;
  return DVAE_OK;
}
DVAE_API int dvae_get_deterministic(void) { return g_dvae_deterministic; }

DVAE_API int dvae_gemm_f32(const void* A, const void* B, void* C, const float* bias, int M, int N, int K,
                           int64_t lda, int64_t ldb, int64_t ldc, int a_kcontig, int b_kcontig, int act,
                           int epi, int split_k, int mode, void* stream) {
  GemmParams p{};
  p.A = A; p.B = B; p.C = C; p.bias = bias;
  p.M = M; p.N = N; p.K = K;
  p.lda = lda; p.ldb = ldb; p.ldc = ldc;
  p.taps = 1; p.tap_mode = 0;
  p.split_k = split_k; p.act = act; p.epi = epi;
  return launch_gemm(p, a_kcontig != 0, b_kcontig != 0, mode, (hipStream_t)stream);
}

DVAE_API int dvae_gemm_f32_batched(const void* const* A, const void* const* B, void* const* C, int batch, int M, int N,
                                   int K, int64_t lda, int64_t ldb, int64_t ldc, int a_kcontig, int b_kcontig, int epi,
                                   int split_k, int mode, void* stream) {
  if (!A || !B || !C || batch < 1 || batch > 4) return DVAE_EINVAL;
  GemmParams p{};
  p.A = A[0]; p.B = B[0]; p.C = C[0]; p.bias = nullptr;
  p.M = M; p.N = N; p.K = K;
  p.lda = lda; p.ldb = ldb; p.ldc = ldc;
  p.taps = 1; p.tap_mode = 0;
  p.split_k = split_k; p.act = DVAE_ACT_NONE; p.epi = epi;
  p.batch = batch;
  for (int b = 0; b < batch; ++b) {
    if (!A[b] || !B[b] || !C[b] || ((((uintptr_t)A[b]) | ((uintptr_t)B[b]) | ((uintptr_t)C[b])) & 15)) return DVAE_EINVAL;
    p.a_boff[b] = (const char*)A[b] - (const char*)A[0];
    p.b_boff[b] = (const char*)B[b] - (const char*)B[0];
    p.c_boff[b] = (char*)C[b] - (char*)C[0];
  }
  return launch_gemm(p, a_kcontig != 0, b_kcontig != 0, mode, (hipStream_t)stream);
}

// ---- k-split WITHOUT atomics (round 6).  When the launch is split (return value n > 1) EVERY split ks stores its partial product
// plainly into slab + ks * slab_stride and C is NOT written; an unsplit launch (n == 1) writes C as `epi` says.  Returns the
// number of k-splits launched (the dispatch may take fewer or — up to slab_cap — more than `split_k`), or a negative error
// code.  The caller combines the n slabs in the fixed order ks = 0, 1, ...: dvae_slab_sum (C = or += their sum),
// dvae_slab_fold.  One writer per element and one summation order: results are run-to-run bit-identical.
DVAE_API int dvae_gemm_f32_slabs(const void* A, const void* B, void* C, float* slab, int64_t slab_stride, int slab_cap,
                                 const float* bias, int M, int N, int K, int64_t lda, int64_t ldb, int64_t ldc,
                                 int a_kcontig, int b_kcontig, int epi, int split_k, int mode, void* stream) {
  if (!slab && split_k > 1) return DVAE_EINVAL;
  GemmParams p{};
  p.A = A; p.B = B; p.C = C; p.bias = bias;
  p.M = M; p.N = N; p.K = K;
  p.lda = lda; p.ldb = ldb; p.ldc = ldc;
  p.taps = 1; p.tap_mode = 0;
  p.split_k = split_k; p.act = DVAE_ACT_NONE; p.epi = epi;
  p.slab = slab; p.slab_stride = slab_stride; p.slab_cap = slab_cap;
  if (slab && slab_stride < (int64_t)M * ldc) return DVAE_EINVAL;
  const int rc = launch_gemm(p, a_kcontig != 0, b_kcontig != 0, mode, (hipStream_t)stream);
  return rc == DVAE_OK ? p.split_k : rc;
}

// batched form: product b stores its split ks into slab (b * n + ks), n = the returned split count (> 1)
DVAE_API int dvae_gemm_f32_batched_slabs(const void* const* A, const void* const* B, void* const* C, int batch, float* slab,
                                         int64_t slab_stride, int slab_cap, int M, int N, int K, int64_t lda, int64_t ldb,
                                         int64_t ldc, int a_kcontig, int b_kcontig, int epi, int split_k, int mode,
                                         void* stream) {
  if (!A || !B || !C || batch < 1 || batch > 4 || (!slab && split_k > 1)) return DVAE_EINVAL;
  GemmParams p{};
  p.A = A[0]; p.B = B[0]; p.C = C[0]; p.bias = nullptr;
  p.M = M; p.N = N; p.K = K;
  p.lda = lda; p.ldb = ldb; p.ldc = ldc;
  p.taps = 1; p.tap_mode = 0;
  p.split_k = split_k; p.act = DVAE_ACT_NONE; p.epi = epi;
  p.batch = batch;
  p.slab = slab; p.slab_stride = slab_stride; p.slab_cap = slab_cap / batch;      // the slabs are shared out evenly
  if (slab && slab_stride < (int64_t)M * ldc) return DVAE_EINVAL;
  for (int b = 0; b < batch; ++b) {
    if (!A[b] || !B[b] || !C[b] || ((((uintptr_t)A[b]) | ((uintptr_t)B[b]) | ((uintptr_t)C[b])) & 15)) return DVAE_EINVAL;
    p.a_boff[b] = (const char*)A[b] - (const char*)A[0];
    p.b_boff[b] = (const char*)B[b] - (const char*)B[0];
    p.c_boff[b] = (char*)C[b] - (char*)C[0];
  }
  const int rc = launch_gemm(p, a_kcontig != 0, b_kcontig != 0, mode, (hipStream_t)stream);
  return rc == DVAE_OK ? p.split_k : rc;
}

// conv forward that also leaves the BatchNorm partial statistics of Y in `bn_ws` (>= dvae_bn_ws_bytes(R, Cout, G) bytes,
// the layout dvae_bn_stats_finalize reads); G = statistics groups (1 or 2)
DVAE_API int dvae_zero_f32(float* x, int64_t n, void* stream);   // elem.hip

namespace {
// A conv (tap mode 1) with FEW output columns (the 80 mel channels: postnet's last conv forward, the data gradient of the
// convs that read the mel) has R / 256 = 64 tall tiles and nothing to fill the chip with: 115 us at 58 TFLOP/s on the
// 128 x 64 tiles.  In the split arithmetic it is cut along k instead (each workgroup all five taps of a quarter of the
// input channels) and accumulated atomically into a result this call clears first: 256 tall workgroups of 40 k-steps.
// Not in the deterministic mode (one writer per element there).
void narrow_conv_split(GemmParams& p, int mode, hipStream_t s) {
  int m = mode;
  if (m >= 0) m &= 0xff;
  if (m == DVAE_MODE_DEFAULT) m = g_dvae_compute_mode;
  if (m != DVAE_MODE_F32X3 || (g_dvae_deterministic && !p.slab) || (mode >= 0 && (mode & ~0xff))) return;
  if (p.N <= 64 || p.N > 128 || (p.N & 3) || p.M < 256 || (((uintptr_t)p.C) & 15)) return;
  const int tiles = (p.M + 255) / 256;
  if (tiles >= 192) return;
  for (int sk = 2; sk <= 8; sk *= 2) {
    if (p.K % (16 * sk)) return;
    const int steps = (p.K / sk / 16) * p.taps;
    if (steps < 24) return;
    if (tiles * sk >= 192) {
      if (p.slab) {      // plain stores into the caller's slabs (dvae_conv5_fwd_slabs / dgrad_t_slabs): the caller sums them
        if (sk > p.slab_cap) return;
        p.split_k = sk;
        return;
      }
      if (dvae_zero_f32((float*)p.C, (int64_t)p.M * p.ldc, s) != DVAE_OK) return;
      p.split_k = sk;
      p.epi = DVAE_EPI_ATOMIC;
      return;
    }
  }
}
}  // namespace

DVAE_API int dvae_conv5_fwd_stats(const void* X, const void* Wp, const float* bias, float* Y, int R, int N, int Cin,
                                  int Cout, int mode, int G, void* bn_ws, void* stream) {
  if (!bn_ws || G < 1 || G > 2 || N < 1 || (N % G) || (R % N)) return DVAE_EINVAL;
  GemmParams p{};
  p.A = X; p.B = Wp; p.C = Y; p.bias = bias;
  p.M = R; p.N = Cout; p.K = Cin;
  p.lda = Cin; p.ldb = Cin; p.ldc = Cout;
  p.taps = 5; p.tap_mode = 1;
  p.a_row_shift = N; p.b_tap_stride = (int64_t)Cout * Cin;
  p.split_k = 1; p.act = DVAE_ACT_NONE; p.epi = DVAE_EPI_STORE;
  p.bn_part = (double*)bn_ws; p.bn_groups = G; p.bn_nseg = N;
  return launch_gemm(p, true, true, mode, (hipStream_t)stream);
}

DVAE_API int dvae_conv5_fwd(const void* X, const void* Wp, const float* bias, float* Y, int R, int N,
                            int Cin, int Cout, int mode, void* stream) {
  GemmParams p{};
  p.A = X; p.B = Wp; p.C = Y; p.bias = bias;
  p.M = R; p.N = Cout; p.K = Cin;
  p.lda = Cin; p.ldb = Cin; p.ldc = Cout;
  p.taps = 5; p.tap_mode = 1;
  p.a_row_shift = N; p.b_tap_stride = (int64_t)Cout * Cin;
  p.split_k = 1; p.act = DVAE_ACT_NONE; p.epi = DVAE_EPI_STORE;
  narrow_conv_split(p, mode, (hipStream_t)stream);
  return launch_gemm(p, true, true, mode, (hipStream_t)stream);
}

// data gradient: the weights packed as Wpt[5][Cin][Cout] (dvae_conv_pack_wt): both operands k-contiguous
DVAE_API int dvae_conv5_dgrad_t(const void* dY, const void* Wpt, float* dX, int R, int N, int Cin, int Cout,
                                int mode, void* stream) {
  GemmParams p{};
  p.A = dY; p.B = Wpt; p.C = dX; p.bias = nullptr;
  p.M = R; p.N = Cin; p.K = Cout;
  p.lda = Cout; p.ldb = Cout; p.ldc = Cin;
  p.taps = 5; p.tap_mode = 1;
  p.a_row_shift = -(int64_t)N; p.b_tap_stride = (int64_t)Cout * Cin;
  p.split_k = 1; p.act = DVAE_ACT_NONE; p.epi = DVAE_EPI_STORE;
  narrow_conv_split(p, mode, (hipStream_t)stream);
  return launch_gemm(p, true, true, mode, (hipStream_t)stream);
}

DVAE_API int dvae_conv5_wgrad(const void* dY, const void* X, float* dWp, int R, int N, int Cin, int Cout,
                              int split_k, int mode, void* stream) {
  GemmParams p{};
  p.A = dY; p.B = X; p.C = dWp; p.bias = nullptr;
  p.M = Cout; p.N = Cin; p.K = R;
  p.lda = Cout; p.ldb = Cin; p.ldc = Cin;
  p.taps = 5; p.tap_mode = 2;
  p.bk_row_shift = N; p.c_tap_stride = (int64_t)Cout * Cin;
  p.split_k = split_k; p.act = DVAE_ACT_NONE; p.epi = DVAE_EPI_ATOMIC;
  return launch_gemm(p, false, false, mode, (hipStream_t)stream);
}

// ---- the three conv products with k-splits into slabs instead of atomics (see dvae_gemm_f32_slabs); return the split count
DVAE_API int dvae_conv5_fwd_slabs(const void* X, const void* Wp, const float* bias, float* Y, float* slab, int64_t slab_stride,
                                  int slab_cap, int R, int N, int Cin, int Cout, int mode, void* stream) {
  GemmParams p{};
  p.A = X; p.B = Wp; p.C = Y; p.bias = bias;
  p.M = R; p.N = Cout; p.K = Cin;
  p.lda = Cin; p.ldb = Cin; p.ldc = Cout;
  p.taps = 5; p.tap_mode = 1;
  p.a_row_shift = N; p.b_tap_stride = (int64_t)Cout * Cin;
  p.split_k = 1; p.act = DVAE_ACT_NONE; p.epi = DVAE_EPI_STORE;
  p.slab = slab; p.slab_stride = slab_stride; p.slab_cap = slab_cap;
  if (slab && slab_stride < (int64_t)R * Cout) return DVAE_EINVAL;
  narrow_conv_split(p, mode, (hipStream_t)stream);
  if (p.split_k == 1) p.slab = nullptr;
  const int rc = launch_gemm(p, true, true, mode, (hipStream_t)stream);
  return rc == DVAE_OK ? p.split_k : rc;
}

DVAE_API int dvae_conv5_dgrad_t_slabs(const void* dY, const void* Wpt, float* dX, float* slab, int64_t slab_stride,
                                      int slab_cap, int R, int N, int Cin, int Cout, int mode, void* stream) {
  GemmParams p{};
  p.A = dY; p.B = Wpt; p.C = dX; p.bias = nullptr;
  p.M = R; p.N = Cin; p.K = Cout;
  p.lda = Cout; p.ldb = Cout; p.ldc = Cin;
  p.taps = 5; p.tap_mode = 1;
  p.a_row_shift = -(int64_t)N; p.b_tap_stride = (int64_t)Cout * Cin;
  p.split_k = 1; p.act = DVAE_ACT_NONE; p.epi = DVAE_EPI_STORE;
  p.slab = slab; p.slab_stride = slab_stride; p.slab_cap = slab_cap;
  if (slab && slab_stride < (int64_t)R * Cin) return DVAE_EINVAL;
  narrow_conv_split(p, mode, (hipStream_t)stream);
  if (p.split_k == 1) p.slab = nullptr;
  const int rc = launch_gemm(p, true, true, mode, (hipStream_t)stream);
  return rc == DVAE_OK ? p.split_k : rc;
}

DVAE_API int dvae_conv5_wgrad_slabs(const void* dY, const void* X, float* dWp, float* slab, int64_t slab_stride, int slab_cap,
                                    int R, int N, int Cin, int Cout, int epi, int split_k, int mode, void* stream) {
  if (!slab && split_k > 1) return DVAE_EINVAL;
  GemmParams p{};
  p.A = dY; p.B = X; p.C = dWp; p.bias = nullptr;
  p.M = Cout; p.N = Cin; p.K = R;
  p.lda = Cout; p.ldb = Cin; p.ldc = Cin;
  p.taps = 5; p.tap_mode = 2;
  p.bk_row_shift = N; p.c_tap_stride = (int64_t)Cout * Cin;
  p.split_k = split_k; p.act = DVAE_ACT_NONE; p.epi = epi;
  p.slab = slab; p.slab_stride = slab_stride; p.slab_cap = slab_cap;
  if (slab && slab_stride < 5 * (int64_t)Cout * Cin) return DVAE_EINVAL;
  const int rc = launch_gemm(p, false, false, mode, (hipStream_t)stream);
  return rc == DVAE_OK ? p.split_k : rc;
}
