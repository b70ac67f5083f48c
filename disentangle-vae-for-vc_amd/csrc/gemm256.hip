// bf16 contraction, 256 x 256 x 64 tile, operands staged by LDS-DMA (round 6).
//
// Replaces aten::addmm / aten::convolution(_backward) of the reference's Conv1d / LSTM / Linear layers
// (/root/reference/model/disentangled_vae.py:111-114,163,172,193) in the bf16 compute mode (BASELINE configs[2], [4]) for
// the products whose operands are both bf16 in memory and that still fill the chip with 256 x 256 tiles.
//
// What differs from gemm_bf16_tall_kernel (256 x 128, buffer_load_dwordx4 -> VGPR -> ds_write_b128):
//   * the operands never pass through registers: `buffer_load_dwordx4 ... lds` (raw_ptr_buffer_load_lds, 16 bytes per
//     lane) writes a wave's 1 KiB piece straight into the LDS image.  No staging VGPRs, no ds_write, and the k loop's
//     memory-instruction stream is 8 LDS-DMA + 24 ds_read_b128 per 32 MFMAs (the tall kernel: 12 loads + 12 ds_write +
//     24 ds_read per 32 MFMAs) — DESIGN.md §9 measured that stream, not the arithmetic, to be what its loop pays for;
//   * one workgroup of EIGHT waves per CU (two per SIMD), each wave a 128 x 64 output tile (4 x 2 MFMA tiles of 32 x 32):
//     L2 -> LDS bytes per MFMA fall by a third.  Two waves per SIMD because an LDS-DMA piece costs its issuing wave
//     100+ cycles of issue (MI355X_MICROARCH.md): with one wave per SIMD (four waves of 128 x 128, measured first:
//     scripts/g256_check.py, 0.55 of the MFMA stream on 8192^3, slower than the tall kernel everywhere) the matrix pipe
//     idles behind every piece; with two the partner's MFMAs run meanwhile;
//   * out-of-range rows, conv padding (negative / past-the-end tap shifts), ragged edges and the tiles past the end are
//     offsets outside the buffer descriptor: the DMA writes ZEROS for them (scripts/probes/glds_probe.hip, measured on
//     MI355X: OOB lanes -> 0 in LDS, destinations above 64 KiB fine).
//
// LDS images (128 KiB: two buffers x {A, B} x 32 KiB; an LDS-DMA piece is lane-linear, base + 16 * lane, so the
// bank-conflict swizzle is applied to the per-lane SOURCE address and again on the read — cdna_hip_programming.md rule 21):
//   k-contiguous operand ([rows][K] in memory): [256 rows][128 B]; 16-byte chunk c of row r sits at chunk c ^ ((r >> 1) & 7)
//     (a ds_read_b128 lane group — 16 rows, one logical chunk — then covers all 16 slots of the 256-byte bank row);
//     piece q (0..31) = rows 8q .. 8q+7, lane l -> row 8q + (l >> 3), physical chunk l & 7: every row read as one whole 128-byte line;
//   row-contiguous operand ([K][rows] in memory): [64 k-rows][512 B]; byte b of k-row kr sits at b ^ ((kr & 3) << 6)
//     (the four k-rows of a ds_read_b64_tr_b16 lane group fall into the four 64-byte quadrants of the bank row);
//     piece q = k-rows 2q, 2q+1, lane l -> k-row 2q + (l >> 5), physical chunk l & 31.
// Fragment k order, MFMA shape (v_mfma_f32_32x32x16_bf16) and the (k-tile, tap) walk are those of gemm_bf16_tall_kernel
// and the 128 x 128 kernel: unsplit results are BIT-IDENTICAL to theirs (tests/test_hip_bf16.py).
//
// Pipeline (two LDS buffers, ONE barrier per 64-deep k-tile, one tile of LDS-DMA in flight across it):
//   iteration i (buffer u = i & 1), fragments of (tile i, sub-step 0) already in registers:
//     sub-steps 0-2: 8 MFMAs each on fragment set s & 1; the 6 ds_read_b128 of the next sub-step in their shadows
//     s_waitcnt vmcnt(0) lgkmcnt(0)      this wave's pieces of tile i+1 have landed; its reads of tile i are done
//     s_barrier                          ... everybody's: buffer u^1 is complete, buffer u is dead
//     sub-step 3:    8 MFMAs; fragment reads of (tile i+1, sub-step 0) from buffer u^1; this wave's 8 LDS-DMA pieces
//                    of tile i+2 into buffer u
// A staged buffer is read only behind the barrier that follows the wait that retired it; it is restaged only behind a
// barrier that every wave reaches after its last read of it has returned (lgkmcnt(0)).
#include "gemm_common.h"

namespace {

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4_t;

template <bool A_KC, bool B_KC, bool BNS, bool BNU>
__global__ __launch_bounds__(512) void gemm_bf16_256_kernel(const GemmParams p) {
  constexpr int BM = 256, BN = 256, BK = 64, MTW = 4, NTW = 2;
  constexpr int OPB = 32768;                 // bytes of one operand's k-tile image
  constexpr unsigned OOB = 0xC0000000u;      // an offset no operand reaches (operands < 1 GiB): the DMA writes zeros
  __shared__ __attribute__((aligned(1024))) char lds[4 * OPB];     // [buffer][A | B]

  const int t = threadIdx.x;
  const int lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave >> 2, wn = wave & 3;
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
  const int ntaps_loop = (p.tap_mode == 1) ? p.taps : 1;
  const int n_iters = ntaps_loop * kiters;
  // k-split without atomics: every split stores its own slab (combined later in the fixed order ks = 0, 1, ...: dvae_slab_sum /
  // dvae_slab_fold); the result buffer is written only by an unsplit launch
  const bool to_slab = p.slab != nullptr && p.split_k > 1;
  float* __restrict__ C = (to_slab ? p.slab + (int64_t)ks * p.slab_stride : (float*)p.C) +
                          (p.tap_mode == 2 ? (int64_t)tap_fixed * p.c_tap_stride : 0);

  const int a_rows = A_KC ? p.M : p.K, b_rows = B_KC ? p.N : p.K;
  const __amdgpu_buffer_rsrc_t a_rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, (int)((int64_t)a_rows * p.lda * 2), 0x00020000);
  const int64_t b_bytes = (int64_t)b_rows * p.ldb * 2 * ((p.tap_mode == 1) ? p.taps : 1);
  const __amdgpu_buffer_rsrc_t b_rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.B, 0, (int)b_bytes, 0x00020000);

  // ---- per-lane source offsets of this wave's 4 + 4 pieces (k-tile 0, tap 2)
  unsigned voa[4], vob[4];
  const int64_t b_shift = (p.tap_mode == 2) ? (int64_t)(tap_fixed - 2) * p.bk_row_shift : 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int q = wave * 4 + j;
    if (A_KC) {
      const int r = 8 * q + (lane >> 3), c = (lane & 7) ^ ((r >> 1) & 7);
      voa[j] = (unsigned)(((int64_t)(m0 + r) * p.lda + k_begin + c * 8) * 2);
      if (m0 + r >= p.M) voa[j] = OOB;
    } else {
      const int kr = 2 * q + (lane >> 5), lc = (lane & 31) ^ ((kr & 3) << 2);
      voa[j] = (unsigned)(((int64_t)(k_begin + kr) * p.lda + m0 + lc * 8) * 2);
      if (m0 + lc * 8 >= p.M) voa[j] = OOB;
    }
    if (B_KC) {
      const int r = 8 * q + (lane >> 3), c = (lane & 7) ^ ((r >> 1) & 7);
      vob[j] = (unsigned)(((int64_t)(n0 + r) * p.ldb + k_begin + c * 8) * 2);
      if (n0 + r >= p.N) vob[j] = OOB;
    } else {
      const int kr = 2 * q + (lane >> 5), lc = (lane & 31) ^ ((kr & 3) << 2);
      vob[j] = (unsigned)((((int64_t)(k_begin + kr) + b_shift) * p.ldb + n0 + lc * 8) * 2);
      if (n0 + lc * 8 >= p.N) vob[j] = OOB;
    }
  }
  const int a_tap_step = (p.tap_mode == 1) ? (int)(p.a_row_shift * p.lda * 2) : 0;
  const int b_tap_step = (p.tap_mode == 1) ? (int)(p.b_tap_stride * 2) : 0;
  const int a_k_step = A_KC ? BK * 2 : (int)(BK * p.lda * 2);
  const int b_k_step = B_KC ? BK * 2 : (int)(BK * p.ldb * 2);
  int tap_n = 0, kit_n = 0;       // next tile to fetch: taps fastest, as the tall kernel walks them
  unsigned a_s = 0, b_s = 0;      // its uniform byte offsets
  auto next_tile_offsets = [&]() {
    const bool live = kit_n < kiters;
    a_s = live ? (unsigned)((tap_n - 2) * a_tap_step + kit_n * a_k_step) : OOB;
    b_s = live ? (unsigned)(tap_n * b_tap_step + kit_n * b_k_step) : OOB;
    const bool wrap = (tap_n + 1 == ntaps_loop);
    tap_n = wrap ? 0 : tap_n + 1;
    kit_n += wrap ? 1 : 0;
  };
  // piece j of this wave (0-3: A, 4-7: B) of the tile whose offsets are current -> buffer `buf`
  auto dma_piece = [&](int buf, int j) {
    char* dst = lds + buf * 2 * OPB + (j >= 4 ? OPB : 0) + (wave * 4 + (j & 3)) * 1024;
    if (j < 4) __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rs, (lds_void_t*)dst, 16, voa[j] + a_s, 0, 0, 0);
    else __builtin_amdgcn_raw_ptr_buffer_load_lds(b_rs, (lds_void_t*)dst, 16, vob[j - 4] + b_s, 0, 0, 0);
  };

  // ---- per-lane fragment addresses (bytes inside one operand image); everything else is an immediate
  // k-contiguous: [sub] -> row (first row of the wave + l31), chunk (2 sub + kh) ^ ((l31 >> 1) & 7); MFMA tile adds 32 rows = 4096 B
  // row-contiguous: [tile] -> k-row tr_k, byte ((first column of the wave + 32 tile + tr_r) * 2) ^ ((tr_k & 3) << 6);
  //                 sub-step adds 16 k-rows = 8192 B, the upper half of the fragment 4 k-rows = 2048 B
  const int g16 = lane >> 4, li = lane & 15;
  const int tr_k = 8 * (g16 >> 1) + (li >> 2), tr_r = 16 * (g16 & 1) + 4 * (li & 3);
  int fa[4], fb[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (A_KC) fa[i] = (wm * 128 + l31) * 128 + (((2 * i + kh) ^ ((l31 >> 1) & 7)) << 4);
    else fa[i] = tr_k * 512 + (((wm * 128 + i * 32 + tr_r) * 2) ^ ((tr_k & 3) << 6));
    if (B_KC) fb[i] = (wn * 64 + l31) * 128 + (((2 * i + kh) ^ ((l31 >> 1) & 7)) << 4);
    else fb[i] = tr_k * 512 + (((wn * 64 + (i & 1) * 32 + tr_r) * 2) ^ ((tr_k & 3) << 6));
  }
  // k-contiguous operand: ds_read_b128 through the LDS array (hipcc waits for it by itself).
  // row-contiguous operand: ds_read_b64_tr_b16 as INLINE ASM.  Through the builtin, hipcc (ROCm 7.2) drains `vmcnt(0)` in front
  // of every transpose read that follows an LDS-DMA — it cannot tell the read from the DMA's destination — which serialised
  // the whole staging pipeline (the weight-gradient shapes ran at 0.6 of the tall kernel; seen in the .s, DESIGN.md section
  // 4.1).  An asm statement is invisible to its wait insertion: the reads of a sub-step are waited for by hand
  // (`s_waitcnt lgkmcnt(0)` + sched_barrier in front of the next sub-step's first MFMA) and their halves are joined into the
  // MFMA operand only behind that wait.
  const unsigned lds_base = (unsigned)(uintptr_t)(lds_void_t*)lds;
  struct Frags {
    bf16x8 a[MTW], b[NTW];                 // k-contiguous operands
    s16x4 a2[MTW][2], b2[NTW][2];          // row-contiguous operands: the two halves of a fragment
  };
  auto tr_read = [&](s16x4& dst, unsigned addr, auto OFF) {
    constexpr int off = decltype(OFF)::value;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off));
  };
  auto read_frag = [&](Frags& f, auto BUF, auto SUB, auto W) {    // W = 0..5: A tiles, then B tiles
    constexpr int buf = decltype(BUF)::value, sub = decltype(SUB)::value, w = decltype(W)::value;
    if constexpr (w < MTW) {
      if constexpr (A_KC) f.a[w] = *reinterpret_cast<const bf16x8*>(lds + buf * 2 * OPB + fa[sub] + w * 4096);
      else {
        const unsigned ad = lds_base + buf * 2 * OPB + fa[w];
        tr_read(f.a2[w][0], ad, std::integral_constant<int, sub * 8192>{});
        tr_read(f.a2[w][1], ad, std::integral_constant<int, sub * 8192 + 2048>{});
      }
    } else {
      constexpr int t = w - MTW;
      if constexpr (B_KC) f.b[t] = *reinterpret_cast<const bf16x8*>(lds + buf * 2 * OPB + OPB + fb[sub] + t * 4096);
      else {
        const unsigned ad = lds_base + buf * 2 * OPB + OPB + fb[t];
        tr_read(f.b2[t][0], ad, std::integral_constant<int, sub * 8192>{});
        tr_read(f.b2[t][1], ad, std::integral_constant<int, sub * 8192 + 2048>{});
      }
    }
  };
  auto join = [](const s16x4& lo, const s16x4& hi) -> bf16x8 {
    return __builtin_shufflevector(__builtin_bit_cast(bf16x4, lo), __builtin_bit_cast(bf16x4, hi), 0, 1, 2, 3, 4, 5, 6, 7);
  };
  auto op_a = [&](const Frags& f, int mt) -> bf16x8 { if constexpr (A_KC) return f.a[mt]; else return join(f.a2[mt][0], f.a2[mt][1]); };
  auto op_b = [&](const Frags& f, int nt) -> bf16x8 { if constexpr (B_KC) return f.b[nt]; else return join(f.b2[nt][0], f.b2[nt][1]); };
  constexpr bool ASM_READS = !A_KC || !B_KC;

  f32x16 acc[MTW][NTW];
#pragma unroll
  for (int i = 0; i < MTW; ++i)
#pragma unroll
    for (int j = 0; j < NTW; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  Frags fr[2];
  // ---- prologue: tiles 0 and 1 in flight, tile 0 waited for, fragments of (tile 0, sub-step 0) read
  next_tile_offsets();
#pragma unroll
  for (int j = 0; j < 8; ++j) dma_piece(0, j);
  next_tile_offsets();
#pragma unroll
  for (int j = 0; j < 8; ++j) dma_piece(1, j);
  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
  for_seq([&](auto W) { read_frag(fr[0], std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, W); },
          std::make_integer_sequence<int, MTW + NTW>{});

  // one k-tile = 32 pinned steps: MFMA g, then at most one fragment read and one LDS-DMA piece
  auto step = [&](auto G, auto U) {
    constexpr int g = decltype(G)::value, u = decltype(U)::value;    // u = parity of the k-tile being computed
    constexpr int sub = g / 8, e = g % 8, mt = e / 2, nt = e % 2;
    if constexpr (ASM_READS && e == 0) {      // the asm transpose reads of this sub-step's fragments (issued a sub-step ago)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
    }
    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(op_a(fr[sub & 1], mt), op_b(fr[sub & 1], nt), acc[mt][nt], 0, 0, 0);
    if constexpr (e < MTW + NTW) {
      if constexpr (sub < 3)
        read_frag(fr[(sub + 1) & 1], std::integral_constant<int, u>{}, std::integral_constant<int, sub + 1>{},
                  std::integral_constant<int, e>{});
      else
        read_frag(fr[0], std::integral_constant<int, u ^ 1>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, e>{});
    }
    if constexpr (sub == 3) dma_piece(u, e);      // tile i+2 -> the buffer tile i just left
    __builtin_amdgcn_sched_barrier(0);
  };
  auto run = [&](auto U) {
    for_seq([&](auto G) { step(G, U); }, std::make_integer_sequence<int, 24>{});
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    for_seq([&](auto G) { step(std::integral_constant<int, 24 + decltype(G)::value>{}, U); }, std::make_integer_sequence<int, 8>{});
  };
  for (int it0 = 0; it0 < n_iters; it0 += 2) {
    next_tile_offsets();      // tile it0 + 2
    run(std::integral_constant<int, 0>{});
    next_tile_offsets();
    run(std::integral_constant<int, 1>{});
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the (all-zero) pieces past the end: nothing may be in flight into LDS at exit

  // ---- epilogue (the tall kernels'): C/D lane map of the 32x32 tile: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5);
  // 16-byte stores after a 4 x 4 transpose inside each lane quad; one straight-line copy per (kind, activation)
  const bool add_bias = (p.bias != nullptr) && (ks == 0);
  auto emit = [&](auto EPI_, auto ACT_) {
    constexpr int epi = decltype(EPI_)::value, act = decltype(ACT_)::value;
    if constexpr (epi == DVAE_EPI_ATOMIC) {
      if (add_bias) {      // bias into the accumulators first (a load pending beside the atomics serialises them)
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
          const int q4 = lane & 3;
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            float x[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const int r = 4 * g + e, dr = e + 8 * g;
              const bool ok = row0 + dr < p.M;
              float v = acc[mt][nt][r] + bias_v;
              v = act_apply(v, act);
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
          // 256 registers per lane here (two waves per SIMD): left alone, hipcc hoists the addresses of a whole tile column
          // and spills 64 accumulator registers — and a scratch reload pending beside the atomics (both count in vmcnt)
          // makes every atomic wait for the one before it (measured: the k-split products at 0.55 of the tall kernel)
          __builtin_amdgcn_sched_barrier(0);
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
  // Plain stores and read-modify-write only: no atomic epilogue here (k-splits go to slabs, see slab / slab_stride) and no
  // tanh (no caller at these sizes).  With both compiled in, the 256 registers a lane has at two waves per SIMD did not hold
  // the epilogue: 64 accumulator registers were spilled in front of it, and scratch reloads pending beside atomics (both
  // count in vmcnt) serialised the atomics — the k-split products ran at 0.55 of the tall kernel.
  using std::integral_constant;
  const int epi_here = to_slab ? DVAE_EPI_STORE : p.epi;      // a split stores its slab plainly
  if (epi_here == DVAE_EPI_ACCUM) emit(integral_constant<int, DVAE_EPI_ACCUM>{}, integral_constant<int, DVAE_ACT_NONE>{});
  else if (p.act == DVAE_ACT_RELU) emit(integral_constant<int, DVAE_EPI_STORE>{}, integral_constant<int, DVAE_ACT_RELU>{});
  else emit(integral_constant<int, DVAE_EPI_STORE>{}, integral_constant<int, DVAE_ACT_NONE>{});
}


#ifdef DVAE_DEV
// ---- experiment (dev build, DVAE_GEMM_256_SHAPE=16): the same pipeline on v_mfma_f32_16x16x32_bf16 — MI355X_MICROARCH.md
// measures 1.12-1.15 x the FLOP/s of the 32x32x16 shape at equal cycles (the power-limited chip holds a higher clock on it).
// k-contiguous operands, plain / bias / ReLU store epilogue only.  The LDS images and the DMA are the shipped kernel's (the
// chunk swizzle c ^ ((row >> 1) & 7) is conflict-free for this shape's ds_read_b128 lane groups too: rows 0-3, 12-15 of one
// chunk and rows 4-11 of the next cover the 16 slots of the bank row).  Results are NOT bit-identical to the other kernels
// (32 k per MFMA instead of 16: another summation order).
__global__ __launch_bounds__(512) void gemm_bf16_256s_kernel(const GemmParams p) {
  constexpr int BM = 256, BN = 256, BK = 64, MT = 8, NT = 4;
  constexpr int OPB = 32768;
  constexpr unsigned OOB = 0xC0000000u;
  __shared__ __attribute__((aligned(1024))) char lds[4 * OPB];
  typedef float f32x4v __attribute__((ext_vector_type(4)));

  const int t = threadIdx.x;
  const int lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int l15 = lane & 15, lq = lane >> 4;
  int tile_m, tile_n;
  gemm_tile_of(p, tile_m, tile_n);
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int ks = blockIdx.z;
  const int k_begin = ks * p.k_per_split;
  const int k_end = min(p.K, k_begin + p.k_per_split);
  const int kiters = (k_end - k_begin) / BK;
  const int ntaps_loop = (p.tap_mode == 1) ? p.taps : 1;
  const int n_iters = ntaps_loop * kiters;
  float* __restrict__ C = (float*)p.C;
  const __amdgpu_buffer_rsrc_t a_rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, (int)((int64_t)p.M * p.lda * 2), 0x00020000);
  const int64_t b_bytes = (int64_t)p.N * p.ldb * 2 * ((p.tap_mode == 1) ? p.taps : 1);
  const __amdgpu_buffer_rsrc_t b_rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.B, 0, (int)b_bytes, 0x00020000);
  unsigned voa[4], vob[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int q = wave * 4 + j;
    const int r = 8 * q + (lane >> 3), c = (lane & 7) ^ ((r >> 1) & 7);
    voa[j] = (unsigned)(((int64_t)(m0 + r) * p.lda + k_begin + c * 8) * 2);
    if (m0 + r >= p.M) voa[j] = OOB;
    vob[j] = (unsigned)(((int64_t)(n0 + r) * p.ldb + k_begin + c * 8) * 2);
    if (n0 + r >= p.N) vob[j] = OOB;
  }
  const int a_tap_step = (p.tap_mode == 1) ? (int)(p.a_row_shift * p.lda * 2) : 0;
  const int b_tap_step = (p.tap_mode == 1) ? (int)(p.b_tap_stride * 2) : 0;
  int tap_n = 0, kit_n = 0;
  unsigned a_s = 0, b_s = 0;
  auto next_tile_offsets = [&]() {
    const bool live = kit_n < kiters;
    a_s = live ? (unsigned)((tap_n - 2) * a_tap_step + kit_n * BK * 2) : OOB;
    b_s = live ? (unsigned)(tap_n * b_tap_step + kit_n * BK * 2) : OOB;
    const bool wrap = (tap_n + 1 == ntaps_loop);
    tap_n = wrap ? 0 : tap_n + 1;
    kit_n += wrap ? 1 : 0;
  };
  auto dma_piece = [&](int buf, int j) {
    char* dst = lds + buf * 2 * OPB + (j >= 4 ? OPB : 0) + (wave * 4 + (j & 3)) * 1024;
    if (j < 4) __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rs, (lds_void_t*)dst, 16, voa[j] + a_s, 0, 0, 0);
    else __builtin_amdgcn_raw_ptr_buffer_load_lds(b_rs, (lds_void_t*)dst, 16, vob[j - 4] + b_s, 0, 0, 0);
  };
  // fragment addresses: [sub] -> row (first row of the wave + l15), chunk (4 sub + lq) ^ (l15 >> 1); MFMA tile adds 16 rows = 2048 B
  int fa[2], fb[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    fa[i] = (wm * 128 + l15) * 128 + (((4 * i + lq) ^ (l15 >> 1)) << 4);
    fb[i] = (wn * 64 + l15) * 128 + (((4 * i + lq) ^ (l15 >> 1)) << 4);
  }
  f32x4v acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4v{0.f, 0.f, 0.f, 0.f};
  bf16x8 fA[2][4], fB[2][4];       // A: set = step & 1 (four m-tiles of one half); B: set = sub & 1
  auto read_a = [&](int set, int buf, int sub, int mh, int i) {
    fA[set][i] = *reinterpret_cast<const bf16x8*>(lds + buf * 2 * OPB + fa[sub] + (4 * mh + i) * 2048);
  };
  auto read_b = [&](int set, int buf, int sub, int i) {
    fB[set][i] = *reinterpret_cast<const bf16x8*>(lds + buf * 2 * OPB + OPB + fb[sub] + i * 2048);
  };
  next_tile_offsets();
#pragma unroll
  for (int j = 0; j < 8; ++j) dma_piece(0, j);
  next_tile_offsets();
#pragma unroll
  for (int j = 0; j < 8; ++j) dma_piece(1, j);
  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int i = 0; i < 4; ++i) read_a(0, 0, 0, 0, i);
#pragma unroll
  for (int i = 0; i < 4; ++i) read_b(0, 0, 0, i);
  // one k-tile = four steps (k32 half `sub`, m-half `mh`) of 16 MFMAs; the next step's fragments are read in its first gaps
  auto step = [&](auto S, auto E, auto U) {
    constexpr int s = decltype(S)::value, e = decltype(E)::value, u = decltype(U)::value;
    constexpr int sub = s >> 1, mh = s & 1, i = e >> 2, nt = e & 3, mt = 4 * mh + i;
    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fA[s & 1][i], fB[sub & 1][nt], acc[mt][nt], 0, 0, 0);
    if constexpr (e < 4) {
      if constexpr (s < 3) read_a((s + 1) & 1, u, (s + 1) >> 1, (s + 1) & 1, e);
      else read_a(0, u ^ 1, 0, 0, e);
    } else if constexpr (e < 8) {
      if constexpr (s == 1) read_b(1, u, 1, e - 4);
      else if constexpr (s == 3) read_b(0, u ^ 1, 0, e - 4);
    }
    if constexpr (s == 3 && (e & 1)) dma_piece(u, e >> 1);
    __builtin_amdgcn_sched_barrier(0);
  };
  auto run = [&](auto U) {
    for_seq([&](auto G) { step(std::integral_constant<int, decltype(G)::value / 16>{},
                               std::integral_constant<int, decltype(G)::value % 16>{}, U); },
            std::make_integer_sequence<int, 48>{});
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    for_seq([&](auto G) { step(std::integral_constant<int, 3>{}, G, U); }, std::make_integer_sequence<int, 16>{});
  };
  for (int it0 = 0; it0 < n_iters; it0 += 2) {
    next_tile_offsets();
    run(std::integral_constant<int, 0>{});
    next_tile_offsets();
    run(std::integral_constant<int, 1>{});
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  // epilogue: C/D of the 16 x 16 tile: col = lane & 15, row = 4 (lane >> 4) + reg; 4 x 4 transpose inside each lane quad, 16-byte stores
  const bool add_bias = p.bias != nullptr;
  const int q4 = lane & 3;
  const bool relu = p.act == DVAE_ACT_RELU;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int col = n0 + wn * 64 + nt * 16 + l15;
      if (col >= p.N) continue;
      const float bv = add_bias ? p.bias[col] : 0.f;
      float x[4], y[4], z[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float v = acc[mt][nt][e] + bv;
        x[e] = relu ? (v > 0.f ? v : 0.f) : v;
      }
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
      const int row = m0 + wm * 128 + mt * 16 + 4 * lq + q4;
      if (row < p.M) *reinterpret_cast<f32x4*>(C + (int64_t)row * p.ldc + (col - q4)) = f32x4{z[0], z[1], z[2], z[3]};
    }
}
#endif

}  // namespace

// launch_gemm (gemm.hip) decides; `p` arrives complete (tiles_m for 256-row tiles, the XCD map, k_per_split a multiple of 64)
int dvae_launch_gemm_bf16_256(const void* params, int a_kc, int b_kc, int bn_uniform, unsigned gx, unsigned gz, hipStream_t s) {
  const GemmParams& p = *static_cast<const GemmParams*>(params);
  const dim3 grid(gx, 1, gz);
#define G256(AK_, BK_, BNS_, BNU_) hipLaunchKernelGGL((gemm_bf16_256_kernel<AK_, BK_, BNS_, BNU_>), grid, dim3(512), 0, s, p)
  if (p.bn_part) {
    if (!a_kc || !b_kc) return DVAE_EINVAL;
    if (bn_uniform) G256(true, true, true, true); else G256(true, true, true, false);
  } else if (a_kc && b_kc) {
#ifdef DVAE_DEV
    if (dvae_dev_knob("DVAE_GEMM_256_SHAPE", 32) == 16 && p.epi == DVAE_EPI_STORE && p.split_k == 1 && p.tap_mode != 2) {
      hipLaunchKernelGGL(gemm_bf16_256s_kernel, grid, dim3(512), 0, s, p);
      return DVAE_OK;
    }
#endif
    G256(true, true, false, false);
  }
  else if (!a_kc && !b_kc) G256(false, false, false, false);
  else return DVAE_EINVAL;      // mixed layouts: no caller at these sizes (launch_gemm keeps them on the tall kernel)
#undef G256
  return DVAE_OK;
}
