// Shared by the contraction translation units (gemm.hip, gemm256.hip): the launch descriptor, the workgroup -> tile map and
// the compile-time step sequencer.  Everything here has internal linkage (each TU compiles its own copy).
#pragma once
#include <type_traits>
#include <utility>
#include "common.h"

namespace {

// Workgroup = WG x WG waves (WG = 2: 128 x 64*NTW tile, 256 threads; WG = 4: 256 x 128*NTW tile, 1024 threads).
// The large tile halves the global-load instructions per MFMA (measured: loads cost ~15 % of the small tile's time).

struct GemmParams {
  const void* A;         // fp32, or bf16 when a16 (bf16 mode only)
  const void* B;
  void* C;               // fp32, or bf16 when c16 (plain store only)
  const float* bias;
  int a16, b16, c16;
  int M, N, K;
  int64_t lda, ldb, ldc;
  int taps;              // 1 or 5
  int tap_mode;          // 0 none | 1 loop over taps, one output | 2 one output per tap (grid.z)
  int64_t a_row_shift;   // mode 1: A row offset per (tap-2)
  int64_t b_tap_stride;  // mode 1: elements between per-tap B matrices
  int64_t bk_row_shift;  // mode 2: B k-row offset per (tap-2)
  int64_t c_tap_stride;  // mode 2: elements between per-tap C matrices
  int split_k;
  int k_per_split;       // multiple of the k-tile
  int act, epi;
  int tiles_m;
  int xcd_map;           // XCD-aware workgroup -> tile map (gemm_tile_of)
  int map_gm, map_gn;    // ... its block of tiles that run on one XCD at a time: map_gm m-tiles x map_gn n-tiles
  int map_nstr;          // ... n-strips of map_gn tiles (tiles_n / map_gn)
  // conv forward feeding a training-mode BatchNorm: per-column partial sums of the stored outputs, one fp64 pair per
  // (64-row chunk, group, column) in the layout bn.hip's finalize kernels read — drops bn_partial's pass over Y
  double* bn_part;
  int bn_groups, bn_nseg;
  // batched launch (dvae_gemm_f32_batched; 128 x 128 kernel, tap_mode 0): `batch` products of one shape in grid.z, product b
  // on A + a_boff[b] ... (byte offsets) — several small under-filled launches become one that fills the chip
  int batch;
  int64_t a_boff[4], b_boff[4], c_boff[4];
  int c_vec;             // 128 x 128 kernel: C rows can be stored / accumulated 16 bytes at a time (N, ldc multiples of 4, aligned)
  // k-split WITHOUT atomics (round 6): a SPLIT launch (split_k > 1) stores split ks's partial product plainly into
  // slab + ks * slab_stride (elements; per-tap outputs at the same c_tap_stride inside a slab) and leaves C alone; whoever
  // needs the result combines the slabs in the fixed order ks = 0, 1, ... (dvae_slab_sum / dvae_slab_fold)
  float* slab;
  int64_t slab_stride;
  int slab_cap;          // slabs the caller provides: at most slab_cap k-splits (host side only)
};

// BF = bf16 compute mode (dvae_set_compute_mode(1); BASELINE configs[2]/[4]): operands stay fp32 in HBM and are rounded
// to bf16 (RNE, v_cvt_pk_bf16_f32) while they are staged into LDS; the products run on v_mfma_f32_32x32x16_bf16 with
// fp32 accumulation.  Images: k-contiguous [rows][BK + 8] bf16 (80-B rows: conflict-free ds_read_b128 of 8 k values);
// row-contiguous [BK][rows + 32] bf16 read with ds_read_b64_tr_b16 (the hardware transpose delivers 4 consecutive k
// of one row per lane; 320-B / 192-B k-rows put the 4 k-rows of a read in 4 different 64-B bank quadrants).
//
// X3 = "fp32 on the bf16 matrix pipe" (dvae_set_compute_mode(2)): the fp32 MFMA runs at the VALU rate (157 TFLOP/s),
// 1/16 of the bf16 MFMA.  Every fp32 operand x is split EXACTLY into three bf16 terms, x = x1 + x2 + x3 (x1 = rne(x),
// x2 = rne(x - x1), x3 = x - x1 - x2: 3 x 8 significand bits + the sign of each residual cover all 24 bits of x), and
// a product a*b is evaluated as the six partial products a1b1 + a1b2 + a2b1 + a1b3 + a2b2 + a3b1, each EXACT in fp32
// (8 x 8 bits), accumulated in fp32 by v_mfma_f32_32x32x16_bf16.  The three dropped terms are <= 2^-24 |ab| together
// — below the rounding of one fp32 FMA — so the result is an fp32 contraction with a different summation order, at
// 6/16 of the fp32-MFMA cost.  The split runs ONCE per element per workgroup, while the k-tile is staged into LDS
// (4.5 VALU operations per element: v_cvt_pk_bf16_f32, two bit operations and a packed subtract per level); LDS holds
// three bf16 images per operand and the fragment reads are those of the bf16 mode, three per tile.  Splitting in
// registers after the fragment read (fp32 images) measured 160 TFLOP/s: every element is then split by both waves
// that read it and the VALU stream sits in front of each MFMA burst.
// A16 / B16M (bf16 mode only): the operand is ALREADY bf16 in memory (activations written as bf16 by their producers,
// bf16 weight copies from the repack launch): half the bytes per element and no conversion on the way into LDS.
// Workgroup -> tile.  Workgroups are dealt round-robin to the 8 XCDs, each with its own 4 MB L2; the workgroups that are
// resident on an XCD at the same time are consecutive in q = blockIdx.x / 8.  With xcd_map, XCD x owns the CONTIGUOUS
// m-tiles [x * tiles_m / 8, (x + 1) * tiles_m / 8) (the +-2-row shifts of the conv taps stay in one L2), and walks its
// part of the tile grid in BLOCKS of map_gm x map_gn tiles — the set that is resident at once — n fastest inside a
// block, the n-strips of one m-group before the next m-group.  The block reads map_gm A tiles + map_gn B tiles from the
// fabric and shares them through the L2 (with one n-tile per round, as before, every A tile went over the fabric once
// per n-tile: 4 GB for the M = 65536, N = 1024, K = 4096 bf16 product, which ran AT the 4 TB/s that allows).
__device__ __forceinline__ void gemm_tile_of(const GemmParams& p, int& tile_m, int& tile_n) {
  if (p.xcd_map) {
    const int per = p.tiles_m >> 3, x = blockIdx.x & 7, q = blockIdx.x >> 3;
    const int blk = p.map_gm * p.map_gn;
    const int round = q / blk, r = q - round * blk;
    const int mg = round / p.map_nstr, st = round - mg * p.map_nstr;
    const int rm = r / p.map_gn;
    tile_m = x * per + mg * p.map_gm + rm;
    tile_n = st * p.map_gn + (r - rm * p.map_gn);
  } else {
    tile_m = blockIdx.x % p.tiles_m;
    tile_n = blockIdx.x / p.tiles_m;
  }
}

template <class F, int... Is>
__device__ __forceinline__ void for_seq(F&& f, std::integer_sequence<int, Is...>) {
  (f(std::integral_constant<int, Is>{}), ...);
}

}  // namespace
