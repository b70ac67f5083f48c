/*
 * dvae_hip.h — C ABI of libdvae_hip.so: the MI355X (gfx950) kernels behind the
 * disentangled-VAE training step.
 *
 * The reference (v-manhlt3/Disentangle-VAE-for-VC) has no FFI/plugin layer: its
 * hot path reaches ATen/cuDNN through torch.nn.  Each entry point below names
 * the reference construct (file:line under /root/reference) whose tensor math it
 * replaces.  INTEGRATION.md shows the ctypes binding a maintainer would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (workspaces too);
 *     the library allocates nothing and keeps no state between calls, except
 *     the opt-in profiling event pool (dvae_prof_*);
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*);
 *     no call synchronises the device;
 *   - return value: 0 on success, a negative DVAE_E* code otherwise; nothing
 *     throws, nothing exits;
 *   - activations are FRAME-MAJOR: a tensor "[T, N, C]" holds frame t of mel
 *     segment n at row r = t*N + n of a row-major [T*N, C] matrix.  N counts
 *     segments of the utterance PAIR batched together (x1 rows first, then x2),
 *     so BatchNorm statistics are taken per GROUP of N/G consecutive segments.
 *   - all arithmetic is fp32 (MFMA v_mfma_f32_32x32x2_f32 / 16x16x4_f32).
 */
#ifndef DVAE_HIP_H
#define DVAE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DVAE_OK 0
#define DVAE_EINVAL (-1)  /* bad shape / alignment / null pointer */
#define DVAE_ELAUNCH (-2) /* hipLaunch failed; see dvae_last_hip_error() */

#define DVAE_ACT_NONE 0
#define DVAE_ACT_RELU 1
#define DVAE_ACT_TANH 2

#define DVAE_EPI_STORE 0  /* C  = result                    */
#define DVAE_EPI_ACCUM 1  /* C += result (plain read-modify-write, no split-K) */
#define DVAE_EPI_ATOMIC 2 /* C += result with global_atomic_add_f32 (split-K allowed) */

/* ABI revision of this header; dvae_version() of the loaded library must return exactly this (the ctypes binding
 * refuses anything else: a stale .so would misread the argument lists below) */
#define DVAE_ABI_VERSION 307
int dvae_version(void);

/* ---- arithmetic of a contraction (every GEMM / conv / LSTM entry point takes a `mode` argument):
 * DVAE_MODE_F32    fp32 operands on v_mfma_f32_32x32x2_f32 / 16x16x4_f32 (the fp32 matrix pipe runs at the vector
 *                  rate, 157 TFLOP/s);
 * DVAE_MODE_F32X3  fp32 RESULTS on the bf16 matrix pipe: each fp32 operand is split exactly into three bf16 terms
 *                  (x = x1 + x2 + x3) and a product is the sum of the six partial products whose weight is >= 2^-16,
 *                  each exact in fp32, accumulated in fp32 — the dropped terms are below the rounding of one fp32 FMA,
 *                  so this IS an fp32 contraction (same error against fp64 as DVAE_MODE_F32, tests/test_hip_x3.py) at
 *                  6/16 of its MFMA cost — BASELINE configs[1], [3];
 * DVAE_MODE_BF16   bf16 operands (rounded to nearest-even on their way into the matrix cores) with fp32 accumulation on
 *                  v_mfma_f32_32x32x16_bf16 / 16x16x32_bf16 — BASELINE configs[2], [4] ("bf16 compute");
 * DVAE_MODE_DEFAULT  whatever dvae_set_compute_mode() last set (process-wide default, initially DVAE_MODE_F32).
 * Everything that is not a contraction (BatchNorm, gates, losses, Adam, master weights) is fp32 in every mode. */
/* bf16 mode only, OR-ed into `mode`: the operand / the result is bf16 IN MEMORY (written so by its producer), not fp32
 * rounded on the way into the matrix cores.  A = first operand (activations), B = second (weights), C = result (plain
 * store epilogue only).  Leading dimensions stay in elements; a bf16 operand needs them (and its contiguous extent)
 * to be multiples of 8. */
#define DVAE_MODE_A_BF16 0x100
#define DVAE_MODE_B_BF16 0x200
#define DVAE_MODE_C_BF16 0x400
#define DVAE_MODE_DEFAULT (-1)
#define DVAE_MODE_F32 0
#define DVAE_MODE_BF16 1
#define DVAE_MODE_F32X3 2
int dvae_set_compute_mode(int mode);
int dvae_get_compute_mode(void);

/* Deterministic mode (a TEST mode; off by default, not on the benchmarked path).  The fast path accumulates split-k
 * partial products, column sums and the recurrences' bias gradients with float atomics, whose order — hence the last
 * bits of every gradient — changes from run to run.  With the mode on every accumulated output element has ONE writer
 * in a fixed order: contractions run unsplit (split_k is forced to 1), dvae_colsum_add walks all rows in one
 * workgroup per column block; callers keep the recurrences' bias gradients out of the persistent launches (ops.py).
 * Two runs of the same step on the same inputs are then bit-identical (tests/test_hip_determinism.py). */
int dvae_set_deterministic(int on);
int dvae_get_deterministic(void);
/* hipError_t of the last failed launch (0 if none) */
int dvae_last_hip_error(void);

/* ---- dense contraction (replaces aten::addmm / mm / convolution[_backward]) ----
 * C[M,N] (+)= act( opA(A)[M,K] * opB(B)[K,N] + bias[N] )
 *   a_kcontig = 1: A stored [M][K] (lda = row stride)   0: stored [K][M]
 *   b_kcontig = 1: B stored [N][K] (torch Linear/LSTM weight layout)   0: stored [K][N]
 * lda, ldb must be multiples of 4 and A, B 16-byte aligned; K % 4 == 0 if an operand is k-contiguous,
 * M % 4 == 0 (N % 4 == 0) if A (B) is not.  A, B, C are fp32 unless `mode` carries DVAE_MODE_A/B/C_BF16 (bf16 mode).
 * Replaces: nn.Linear forward/backward (disentangled_vae.py:165-171,194,211-213,232-233,247),
 * LSTM input projections and weight gradients (:163,172,193).
 */
int dvae_gemm_f32(const void* A, const void* B, void* C, const float* bias,
                  int M, int N, int K, int64_t lda, int64_t ldb, int64_t ldc,
                  int a_kcontig, int b_kcontig, int act, int epi, int split_k, int mode, void* stream);

/* `batch` (1..4) products of ONE shape in one launch: product b is C[b] (+)= opA(A[b]) * opB(B[b]) (no bias, no
 * activation; epi / split_k / mode as above, storage flags apply to every product).  For contractions too small to fill
 * the chip on their own — the weight gradients of the two directions of the H = 64 encoder BiLSTM (disentangled_vae.py:163):
 * dW_ih, dW_hh of the forward and the reverse direction are two launches of batch 2 instead of four under-filled ones.
 * A, B, C: host arrays of `batch` device pointers (read during the call only), each 16-byte aligned. */
int dvae_gemm_f32_batched(const void* const* A, const void* const* B, void* const* C, int batch, int M, int N, int K,
                          int64_t lda, int64_t ldb, int64_t ldc, int a_kcontig, int b_kcontig, int epi, int split_k,
                          int mode, void* stream);
/* ---- k-split WITHOUT atomics (round 6; replaces the split-K atomic accumulation of every aten::mm / convolution_backward
 * the reference runs through cuBLAS / cuDNN: /root/reference/model/variational_base_vae.py:68 `loss.backward()`) ----
 * A product cut along k stores its partial results PLAINLY: when the launch is split (return value n > 1) EVERY split ks
 * stores into slab + ks * slab_stride (elements; a multiple of 4, >= the extent of C; the slabs 16-byte aligned; conv weight
 * gradients keep their five per-tap outputs at the same distances inside a slab) and C is NOT written; an unsplit launch
 * (n == 1) writes C as `epi` says (DVAE_EPI_STORE / DVAE_EPI_ACCUM).  `slab_cap` = slabs the caller provides: the dispatch may
 * take fewer k-splits than `split_k` asks for, or — the tiles that want one workgroup per CU — more, never more than
 * slab_cap.  RETURNS the number of k-splits launched (>= 1), or a negative error code.  The caller then combines the n
 * slabs in the fixed order ks = 0, 1, ... (dvae_slab_sum now, or dvae_slab_fold later, e.g. at the end of the backward pass):
 * every output element has ONE writer per buffer and ONE summation order, so results are run-to-run bit-identical, and every
 * epilogue is plain 16-byte stores (no read-modify-write, no atomics).  The bias rides split 0; no activation when split
 * (dvae_slab_sum applies it).  dvae_conv5_fwd_slabs / dvae_conv5_dgrad_t_slabs: the convs with <= 128 output columns, which
 * the default arithmetic cuts along k to fill the chip (they return 1 when they did not split). */
int dvae_gemm_f32_slabs(const void* A, const void* B, void* C, float* slab, int64_t slab_stride, int slab_cap,
                        const float* bias, int M, int N, int K, int64_t lda, int64_t ldb, int64_t ldc,
                        int a_kcontig, int b_kcontig, int epi, int split_k, int mode, void* stream);
/* batched form (dvae_gemm_f32_batched): product b stores split ks into slab (b * n + ks), n = the return value (> 1) */
int dvae_gemm_f32_batched_slabs(const void* const* A, const void* const* B, void* const* C, int batch, float* slab,
                                int64_t slab_stride, int slab_cap, int M, int N, int K, int64_t lda, int64_t ldb,
                                int64_t ldc, int a_kcontig, int b_kcontig, int epi, int split_k, int mode, void* stream);
int dvae_conv5_fwd_slabs(const void* X, const void* Wp, const float* bias, float* Y, float* slab, int64_t slab_stride,
                         int slab_cap, int R, int N, int Cin, int Cout, int mode, void* stream);
int dvae_conv5_dgrad_t_slabs(const void* dY, const void* Wpt, float* dX, float* slab, int64_t slab_stride, int slab_cap,
                             int R, int N, int Cin, int Cout, int mode, void* stream);
int dvae_conv5_wgrad_slabs(const void* dY, const void* X, float* dWp, float* slab, int64_t slab_stride, int slab_cap,
                           int R, int N, int Cin, int Cout, int epi, int split_k, int mode, void* stream);
/* C[i] = act((accumulate ? C[i] : 0) + sum_{k < nslab} slab[k * slab_stride + i]), i < n  (n, slab_stride multiples of 4) */
int dvae_slab_sum(float* C, const float* slab, int64_t slab_stride, int nslab, int64_t n, int act, int accumulate,
                  void* stream);
/* c[i] += sum_{k < nslab} slab[...] for many results in ONE launch per DVAE_SLAB_FOLD_MAX entries (`descs`: read during the call) */
typedef struct {
  float* c;
  const float* slab;
  int64_t slab_stride;
  int64_t n;
  int nslab;
  int pad_;
} dvae_slab_desc_t;
#define DVAE_SLAB_FOLD_MAX 64
int dvae_slab_fold(const dvae_slab_desc_t* descs, int n_entries, void* stream);
/* ---- Conv1d(k=5, stride 1, pad 2) on frame-major data (disentangled_vae.py:111-114, 178-181) ----
 * Weights are used in PACKED form Wp[5][Cout][Cin] (see dvae_conv_pack_w).
 * fwd : Y[R,Cout]   = sum_tap X[r+(tap-2)*N, :] * Wp[tap]^T + bias      (rows outside [0,R) are zero)
 * dgrad: dX[R,Cin]  = sum_tap dY[r-(tap-2)*N, :] * Wp[tap]                  (dvae_conv5_dgrad_t, on the transposed pack)
 * wgrad: dWp[tap][Cout][Cin] += sum_r dY[r, co] * X[r+(tap-2)*N, ci]     (atomic accumulation)
 * R = T*N rows, N = segments per frame.
 */
int dvae_conv5_fwd(const void* X, const void* Wp, const float* bias, float* Y,
                   int R, int N, int Cin, int Cout, int mode, void* stream);
/* conv forward that also leaves the per-column partial sums of Y that a training-mode BatchNorm over G statistics groups
 * needs in bn_ws (>= dvae_bn_ws_bytes(R, Cout, G) bytes): follow with dvae_bn_stats_finalize instead of
 * dvae_bn_stats_fwd — one pass over Y less per block (disentangled_vae.py:151-162: conv -> BatchNorm -> ReLU) */
int dvae_conv5_fwd_stats(const void* X, const void* Wp, const float* bias, float* Y,
                         int R, int N, int Cin, int Cout, int mode, int G, void* bn_ws, void* stream);
int dvae_conv5_wgrad(const void* dY, const void* X, float* dWp,
                     int R, int N, int Cin, int Cout, int split_k, int mode, void* stream);
/* W[Cout][Cin][5] (torch layout, state_dict contract) -> Wp[5][Cout][Cin] */
int dvae_conv_pack_w(const float* W, float* Wp, int Cout, int Cin, void* stream);
/* Wpt[5][Cin][Cout]: the transposed pack; with it the data gradient reads BOTH operands k-contiguously (the faster
 * ds_read_b128 fragment path of the contraction kernel). */
int dvae_conv_pack_wt(const float* W, float* Wpt, int Cout, int Cin, void* stream);
int dvae_conv5_dgrad_t(const void* dY, const void* Wpt, float* dX, int R, int N, int Cin, int Cout, int mode,
                       void* stream);
/* dW[Cout][Cin][5] += dWp[5][Cout][Cin] */
int dvae_conv_unpack_add_w(const float* dWp, float* dW, int Cout, int Cin, void* stream);

/* ---- BatchNorm1d in training mode + activation (disentangled_vae.py:58,69,78,159,182,189; F.relu :202,243; tanh :83) ----
 * Statistics are per GROUP g = (r % N) / (N/G) (one group per encode()/decode()/postnet() call of the
 * reference, which sees x1 and x2 separately).  ws: >= dvae_bn_ws_bytes(R, C, G) bytes.
 * dvae_bn_stats_fwd: mean[G][C], rstd[G][C] (biased var, eps); running_mean/var updated once per group in
 *   group order with `momentum` and the unbiased variance; *num_batches_tracked += G. running_* may be null.
 * dvae_bn_apply_fwd: Z = act((Y-mean)*rstd*gamma + beta) (+ residual if non-null)
 * dvae_bn_bwd: given dZ, saved Y and Z: dY (may alias dZ), dgamma += , dbeta += ; the residual branch's
 *   gradient is dZ itself (handled by the caller).
 */
int64_t dvae_bn_ws_bytes(int R, int C, int G);
int dvae_bn_stats_fwd(const float* Y, float* mean, float* rstd, float* running_mean, float* running_var,
                      int64_t* num_batches_tracked, void* ws, int R, int N, int C, int G,
                      float eps, float momentum, void* stream);
/* finalize only: the partial sums are already in ws (left there by dvae_conv5_fwd_stats) */
int dvae_bn_stats_finalize(float* mean, float* rstd, float* running_mean, float* running_var,
                           int64_t* num_batches_tracked, const void* ws, int R, int N, int C, int G,
                           float eps, float momentum, void* stream);
int dvae_bn_apply_fwd(const float* Y, const float* mean, const float* rstd, const float* gamma,
                      const float* beta, const float* residual, void* Z,
                      int R, int N, int C, int G, int act, int z_bf16, void* stream);
/* dtypes: bit 0 = Z is bf16, bit 1 = dY is written as bf16 (bf16 compute mode: Z and dY are contraction operands, their
 * producers write them in the storage the consumers read; z_bf16 of dvae_bn_apply_fwd likewise) */
int dvae_bn_bwd(const float* dZ, const float* Y, const void* Z, const float* mean, const float* rstd,
                const float* gamma, void* dY, float* dgamma, float* dbeta, void* ws,
                int R, int N, int C, int G, int act, int dtypes, void* stream);
/* the same without reading Z, for act = ReLU or none: the activation's derivative is recomputed from Y with the forward
 * pass's expression (act((Y-mean)*rstd*gamma + beta) > 0) — 20 instead of 28 bytes per element over the two passes.
 * (tanh blocks keep dvae_bn_bwd: recomputing tanh costs these HBM-bound passes more than the bytes it saves.) */
int dvae_bn_bwd_from_y(const float* dZ, const float* Y, const float* mean, const float* rstd, const float* gamma,
                       const float* beta, void* dY, float* dgamma, float* dbeta, void* ws,
                       int R, int N, int C, int G, int act, int dtypes, void* stream);

/* ---- LSTM recurrence, frame-major, one launch per frame (nn.LSTM at disentangled_vae.py:163,172,193) ----
 * One dvae_lstm_dir_t per direction (1 or 2).  Gate order i,f,g,o (torch).
 * fwd : gates[T,N,4H] holds X*W_ih^T + b_ih + b_hh on entry and the ACTIVATED gates on exit;
 *       h_out[t] (row stride ldh; direction d writes columns [d*H,(d+1)*H) via its own pointer), c_all[T,N,H].
 * bwd : w_hh_t is W_hh transposed, [H][4H]; dh_out = gradient w.r.t. h_out (same strides);
 *       dgates[T,N,4H] = gradient w.r.t. the pre-activation gates; dc_ws[N,H] scratch.
 */
typedef struct {
  float* gates;       /* [T,N,4H] */
  const float* w_hh;  /* fwd: [4H,H]   bwd: W_hh^T [H,4H] */
  float* h_out;       /* fwd: out [T,N,ldh]   bwd: unused */
  float* c_all;       /* [T,N,H] */
  const float* dh_out;/* bwd only: [T,N,ldh] */
  float* dgates;      /* bwd only: [T,N,4H] */
  float* dc_ws;       /* bwd only: [N,H] */
  const float* w_packed; /* optional: weights re-packed in MFMA fragment order by dvae_lstm_pack_w (fwd: packed_fwd,
                            bwd: packed_bwd); when given, the faster 1-KiB-burst frame kernels are used */
  int reverse;        /* 0: t = 0..T-1, 1: t = T-1..0 */
  int packed_mode;    /* DVAE_MODE_F32: w_packed from dvae_lstm_pack_w (fp32 MFMA recurrence); DVAE_MODE_BF16: from
                         dvae_lstm_pack_w_bf16 (bf16 operands / fp32 accumulation on v_mfma_f32_16x16x32_bf16);
                         DVAE_MODE_F32X3: from dvae_lstm_pack_w_x3 (fp32 results: three bf16 planes of W_hh, h split the
                         same way, six exact partial products).  The last two need H % 512 == 0, or H == 64: the whole-sequence
                         kernels of H = 64 take w_hh itself (no w_packed) and round / split it in registers */
  int step_shift;     /* *_range calls: this entry runs its step s in the launch of global step s + step_shift (0 for
                         the plain calls).  Lets two STACKED layers share launches, the upper one a chunk of frames
                         behind the lower one (whose chunk of outputs has meanwhile gone through the upper layer's input
                         projection): twice the workgroups per launch, half the launches */
  int state_bf16;     /* bf16 mode only: h_out (forward call) / dgates (backward call) are bf16 tensors — the storage their
                         consumers (the next frame, the next layer's projection, the weight gradients) read; the same
                         element strides */
  void* pers_ws;      /* optional: >= dvae_lstm_pers_ws_bytes(N, H) bytes, 256-byte aligned, zero-initialised ONCE by the
                         caller and never written by it afterwards: the flags in it count frames across ALL launches (each
                         launch reads the epoch the previous one left, ABI 306: no clearing launch in front of a persistent
                         launch), so a workspace must not be cleared, copied or shared between devices.  When given to a plain one-direction dvae_lstm_seq_fwd / _bwd call whose shape has a
                         persistent kernel (dvae_lstm_pers_supported), the whole sequence
                         runs in ONE W_hh-resident launch (csrc/lstm_pers.hip) instead of one launch per frame; same
                         arithmetic, same tensors.  One workspace serves every layer run on one stream */
  unsigned pers_timeout_us; /* bound of every cross-workgroup wait of that launch (0: 2 s); see dvae_lstm_pers_check */
  float* dbias_ih;    /* optional, used by the PERSISTENT backward launch only (dvae_lstm_pers_supported(.., bwd = 1)): the */
  float* dbias_hh;    /* column sums of dgates over all frames and rows are ADDED to these [4H] vectors (the gradients of
                         b_ih and b_hh, nn.LSTM keeps two): the caller then skips its dvae_colsum_add pass over dgates.
                         Ignored (and the caller must run dvae_colsum_add) when the per-frame kernels are used */
  float* dbias_part;  /* optional, PERSISTENT backward launch only, instead of dbias_ih / dbias_hh (ABI 307): row group rb of the
                         launch STORES its share of the column sums of dgates at dbias_part[(rb * 4 + g) * H + unit] — a
                         [DVAE_PERS_BIAS_SLABS][4H] buffer (row groups the launch does not have are written as zeros) that
                         the caller adds up in the fixed order rb = 0, 1, ... (dvae_slab_sum / dvae_slab_fold with
                         slab_stride 4H): no atomics, run-to-run bit-identical bias gradients */
  int64_t gate_ld;    /* row stride, in elements, of gates and dgates (0: 4H, the dense layout).  H = 64 only: the two
                         directions of the encoder BiLSTM keep their gates side by side in ONE [T*N, 8H] tensor, so that both
                         input projections are one contraction with N = 8H (and both data gradients one with K = 8H);
                         every direction of a call must name the same stride */
} dvae_lstm_dir_t;
#define DVAE_PERS_BIAS_SLABS 16  /* rows of dvae_lstm_dir_t.dbias_part */
/* W_hh [4H,H] -> fragment-ordered copies (each 4H*H floats) for the forward / backward frame kernels */
int dvae_lstm_pack_w(const float* w_hh, float* packed_fwd, float* packed_bwd, int H, void* stream);
/* bf16 compute mode: the same copies rounded to bf16 (each 4H*H bf16 values = 2*4H*H bytes) */
int dvae_lstm_pack_w_bf16(const float* w_hh, void* packed_fwd, void* packed_bwd, int H, void* stream);
/* fp32x3 mode: three bf16 planes per fragment, w == w1 + w2 + w3 exactly (each copy 3 * 4H*H bf16 values = 6*4H*H bytes) */
int dvae_lstm_pack_w_x3(const float* w_hh, void* packed_fwd, void* packed_bwd, int H, void* stream);
/* ---- all weight-derived operand layouts of a model in ONE launch (once per training step, after Adam) ----
 * kind CONV_T    src Wp[5][d0=Cout][d1=Cin]                 -> dst Wpt[5][Cin][Cout]  (operand of dvae_conv5_dgrad_t)
 *      LSTM_PACK src W_hh[4*d0][d0], d0 = H                 -> dst packed_fwd in mode d1, dst2 packed_bwd in mode d2
 *                                                              (either pointer may be null; DVAE_MODE_F32 / _BF16 /
 *                                                              _F32X3, the last two only where H % 512 == 0): see
 *                                                              dvae_lstm_pack_w / _bf16 / _x3
 *      TRANSPOSE src W[d0][d1]                              -> dst W^T[d1][d0]         (bf16 destination when d2 & 1;
 *                                                                                       CONV_T likewise); d2 >> 1 = row
 *                                                                                       stride of dst in elements (0: d0)
 *      COPY_F32  src [d0] fp32                              -> dst, same layout        (weights of two directions side by side)
 *      ADD2      src, src2 [d0]                             -> dst = src + src2        (b_ih + b_hh)
 *      CAST_BF16 src [d0*d1] fp32                           -> dst bf16, same layout   (bf16 mode: weight operands)
 * `descs` is a HOST array of n <= 72 entries (copied into the launch); the device buffers are the caller's. */
#define DVAE_REPACK_CONV_T 0
#define DVAE_REPACK_LSTM_PACK 1
#define DVAE_REPACK_TRANSPOSE 2
#define DVAE_REPACK_ADD2 3
#define DVAE_REPACK_CAST_BF16 4
#define DVAE_REPACK_COPY_F32 5
typedef struct {
  int kind, d0, d1, d2;
  const void* src;
  const void* src2;
  void* dst;
  void* dst2;
} dvae_repack_desc_t;
int dvae_repack_all(const dvae_repack_desc_t* descs, int n, void* stream);

/* ---- W_hh-resident persistent recurrence (nn.LSTM at disentangled_vae.py:172,193; H = 512 / 1024) ----
 * Exists for: DVAE_MODE_BF16, forward and backward, (H/32) * ceil(N/32) <= CU count; DVAE_MODE_F32X3, FORWARD only (three
 * bf16 planes of W_hh resident, h handed over as three planes), N <= 128 (the backward recurrence streams the 4H-wide
 * gate gradients, which no residency shrinks: it stays on the per-frame kernels).
 * dvae_lstm_pers_supported: 1 when a call with (N, H, packed_mode = mode, pass) would take the persistent launch on the
 *   current device (given a workspace), else 0.
 * dvae_lstm_pers_ws_bytes: size of the synchronisation workspace (flags + sticky error record + the flags' epoch + exchange
 *   ring: two slots in use, four sized) for (N, H); 0 when the shape has no persistent kernel.
 * dvae_lstm_pers_check: SYNCHRONISES `stream`, then returns DVAE_ELAUNCH if a bounded wait of any persistent launch on
 *   this workspace gave up since the last check (info4 = {code 1 fwd / 2 bwd, workgroup, step, wave}; the outputs of
 *   that launch are garbage), DVAE_OK otherwise.  Not capturable; call it wherever the host synchronises anyway.
 * dvae_lstm_pers_selftest: a forward launch in which workgroup `drop_bid` never publishes — every waiter must give up
 *   within dir->pers_timeout_us and dvae_lstm_pers_check must then report it (tests/test_hip_lstm_pers.py). */
int dvae_lstm_pers_supported(int N, int H, int mode, int bwd);
int64_t dvae_lstm_pers_ws_bytes(int N, int H);
int dvae_lstm_pers_check(void* ws, int* info4, void* stream);
/* device address of the sticky error word of workspace `ws` (non-zero from the first bounded wait that gave up until
 * dvae_lstm_pers_check reported it): the `skip_if_nonzero` argument of dvae_adam_flat_dev */
const unsigned* dvae_lstm_pers_err_word(void* ws);
int dvae_lstm_pers_selftest(const dvae_lstm_dir_t* dir, int T, int N, int H, int64_t ldh, int drop_bid, void* stream);

int dvae_lstm_seq_fwd(const dvae_lstm_dir_t* dirs, int ndir, int T, int N, int H, int64_t ldh, void* stream);
int dvae_lstm_seq_bwd(const dvae_lstm_dir_t* dirs, int ndir, int T, int N, int H, int64_t ldh, void* stream);
/* launches of global steps [step_begin, step_end) only (H a multiple of 512, packed weights); the `dirs` entries may be
 * two directions of one layer or, with step_shift, two stacked layers of equal H, N, T and ldh */
int dvae_lstm_seq_fwd_range(const dvae_lstm_dir_t* dirs, int ndir, int T, int N, int H, int64_t ldh,
                            int step_begin, int step_end, void* stream);
int dvae_lstm_seq_bwd_range(const dvae_lstm_dir_t* dirs, int ndir, int T, int N, int H, int64_t ldh,
                            int step_begin, int step_end, void* stream);

/* ---- reparameterise + latent assembly (disentangled_vae.py:222-228, 252-272) ----
 * style[N,2S], content[N,2Cn] with N = 2*Bh (x1 rows then x2 rows); eps_c[N,Cn] (null => z = mu), eps_s[Bh,S].
 * Outputs: z[N,S+Cn], q_mu[N,S+Cn], q_lv[N,S+Cn] (rows [0,Bh) = q_z1, [Bh,N) = q_z2), s_mu[Bh,S], s_lv[Bh,S].
 * The x2 style head is detached (:257-258): its gradient rows are written as zero.
 */
int dvae_latent_fwd(const float* style, const float* content, const float* eps_c, const float* eps_s,
                    float* z, float* q_mu, float* q_lv, float* s_mu, float* s_lv,
                    int Bh, int S, int Cn, void* stream);
int dvae_latent_bwd(const float* style, const float* content, const float* eps_c, const float* eps_s,
                    const float* dz, const float* dq_mu, const float* dq_lv, const float* ds_mu,
                    const float* ds_lv, float* dstyle, float* dcontent, int Bh, int S, int Cn, void* stream);

/* ---- KL reduction (disentangled_vae.py:320-323): out[0] = scale * sum(1 + lv - mu^2 - exp(lv)) ---- */
int dvae_kl_fwd(const float* mu, const float* lv, float* out, int64_t n, float scale, void* stream);
/* dmu = g*scale*(-2mu), dlv = g*scale*(1-exp(lv)); g = *gout (device scalar) */
int dvae_kl_bwd(const float* mu, const float* lv, const float* gout, float* dmu, float* dlv,
                int64_t n, float scale, void* stream);

/* ---- L1 reconstruction loss, reduction='sum' (disentangled_vae.py:314-318): out[0] = scale*sum|x-y| ----
 * ws: >= dvae_l1_ws_bytes(n). */
int64_t dvae_l1_ws_bytes(int64_t n);
int dvae_l1_sum_fwd(const float* x, const float* y, float* out, void* ws, int64_t n, float scale, void* stream);
/* dy = -sign(x-y) * g * scale */
int dvae_l1_sum_bwd(const float* x, const float* y, const float* gout, float* dy, int64_t n, float scale,
                    void* stream);

/* ---- the whole loss_functionGVAE2 (disentangled_vae.py:310-327) in two launches forward, one backward ----
 * out8 = (LOSS, L1(x1,recon1), L1(x2,recon2), L1(x1,recon1_hat), L1(x2,recon2_hat), KL_z1, KL_z2, KL_style) with
 *   L1(x, r) = l1_scale * sum|x - r|                      (l1_scale = 1 / batch_size: F.l1_loss(sum).div(batch_size))
 *   KL_zk    = kl_scale * sum(1 + lv - mu^2 - exp(lv))   (kl_scale = -0.5 / rows of q_zk_mu: torch.mean over the batch)
 *   KL_style = style_scale * sum(...)                     (report only)
 *   LOSS     = mse_cof * (((L1_1 + L1_2) + L1_3) + L1_4) + kl_cof * (KL_z1 + KL_z2)
 * bwd: g8 = gradient w.r.t. out8 (a device vector: no host round trip); any output pointer may be null (not needed). */
typedef struct {
  const float *x1, *x2, *recon1, *recon2, *recon1_hat, *recon2_hat;   /* [n] each, 16-byte aligned */
  const float *q1_mu, *q1_lv, *q2_mu, *q2_lv;                          /* [nq] each */
  const float *s_mu, *s_lv;                                            /* [ns] each */
  int64_t n;
  int nq, ns;
  float l1_scale, kl_scale, style_scale, mse_cof, kl_cof;
} dvae_loss_desc_t;
int64_t dvae_loss_ws_bytes(int64_t n);
int dvae_loss_fwd(const dvae_loss_desc_t* desc, float* out8, void* ws, void* stream);
int dvae_loss_bwd(const dvae_loss_desc_t* desc, const float* g8, float* d_recon1, float* d_recon2, float* d_recon1_hat,
                  float* d_recon2_hat, float* d_q1_mu, float* d_q1_lv, float* d_q2_mu, float* d_q2_lv, float* d_s_mu,
                  float* d_s_lv, void* stream);

/* ---- Adam over one flat parameter buffer (torch.optim.Adam at disentangled_vae.py:304) ----
 * g is multiplied by grad_scale first (1/world_size after a sum all-reduce). `step` is 1-based. */
int dvae_adam_flat(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1,
                   float beta2, float eps, float grad_scale, int step, void* stream);

/* Same update with every scalar that changes between steps kept ON THE DEVICE, so that a captured hipGraph replays a
 * correct Adam step and a learning-rate schedule needs no re-capture.  state: float[8], zero-initialised by the caller:
 *   [0] t (advanced by this call)   [1] 1 - beta1^t   [2] sqrt(1 - beta2^t)   [4] lr   [5] grad_scale
 * ([4], [5] written by the caller — optim.FlatAdam.sync_scalars — outside any capture).
 * skip_if_nonzero (optional): a device word; while it is non-zero the call changes NOTHING (no tick, no update, no clear).
 *   optim.FlatAdam passes the sticky error word of the persistent LSTM launches (dvae_lstm_pers_err_word): a launch that gave
 *   up a bounded wait left garbage gradients, and the weights / moments must not consume them before the host has looked.
 * clear: up to 8 ranges [lo, hi) of g (multiples of 4 elements) that are zeroed AFTER they were read — the
 *   optimizer.zero_grad() of the NEXT step (variational_base_vae.py:86) rides on this launch's pass over g.
 * tick: 1 advances t and the bias corrections first (one call per step); 0 applies the update of the CURRENT t to another
 *   slice of the buffers (a sharded optimizer — ddp.GradReducer(mode="rs_ag") — updates one slice per bucket; clear ranges
 *   are relative to the `g` passed).
 * n % 4 == 0. */
typedef struct {
  int64_t lo[8], hi[8];
  int n;
} dvae_ranges_t;
int dvae_adam_flat_dev(float* p, float* g, float* m, float* v, int64_t n, float beta1, float beta2, float eps,
                       float* state, const unsigned* skip_if_nonzero, const dvae_ranges_t* clear, int tick, void* stream);

/* x[0, n) = 0 (16-byte aligned): optimizer.zero_grad() (variational_base_vae.py:86) and the outputs that split-k
 * contractions accumulate into atomically, zeroed by a launch of their own right in front of the accumulation. */
int dvae_zero_f32(float* x, int64_t n, void* stream);

/* out = a + b (+ c when non-null): the sum of the gradients of a tensor with several consumers — x feeding both
 * latent heads (disentangled_vae.py:212-215), the decoder output feeding the postnet, its residual and the loss
 * (variational_base_vae.py:292-293).  n % 4 == 0, 16-byte aligned. */
int dvae_sum_f32(const float* a, const float* b, const float* c, float* out, int64_t n, void* stream);

/* ---- layout plumbing ----
 * dvae_mel_to_frames: x1,x2 [Bh,C,T] (torch layout, variational_base_vae.py:81-82) -> X[T, 2*Bh, C]; x2 may be
 *   null (then N = Bh).  dvae_frames_to_mel is the inverse ([T,N,C] -> out[N,C,T]).
 * dvae_permute_102: in[A,B,C] -> out[B,A,C]
 * dvae_colsum_add: out1[c] += sum_r X[r,c] (and out2 if non-null); X row stride ld; X fp32 or (x_bf16) bf16.
 * dvae_transpose: in[R,C] -> out[C,R]
 */
int dvae_mel_to_frames(const float* x1, const float* x2, void* X, int Bh, int C, int T, int out_bf16, void* stream);
int dvae_frames_to_mel(const float* X, float* out, int N, int C, int T, void* stream);
int dvae_permute_102(const float* in, float* out, int A, int B, int C, void* stream);
int dvae_colsum_add(const void* X, float* out1, float* out2, int R, int C, int64_t ld, int x_bf16, void* stream);
/* the same WITHOUT atomics (round 6): row blocks store partial sums into `ws`, the last workgroup of a column block to arrive
 * adds them up in row-block order and is the one writer of out[c] — run-to-run bit-identical.  ws: >= dvae_colsum_ws_bytes(R, C)
 * bytes, 16-byte aligned, its first 4096 bytes ZERO before the first call (every call leaves them zero); one ws per stream
 * that may have such a launch in flight */
int64_t dvae_colsum_ws_bytes(int R, int C);
int dvae_colsum_add_ws(const void* X, float* out1, float* out2, int R, int C, int64_t ld, int x_bf16, void* ws, void* stream);
int dvae_transpose(const float* in, float* out, int R, int C, void* stream);
/* dU = dZ * act'(Z), Z = activation OUTPUT (ReLU after enc_linear, disentangled_vae.py:211); dU may alias dZ */
/* Y = act(Y) in place (used after a split-K Linear) */
int dvae_act_fwd(float* Y, int64_t n, int act, void* stream);
int dvae_act_bwd(const float* dZ, const float* Z, float* dU, int64_t n, int act, void* stream);

/* ---- input pipeline on the device (SpeechDatasetGVAE.__getitem__, preprocessing/dataset.py:93-114, for a whole batch)
 * mels[n_utt, C, Lmax] padded store, lens[n_utt]; out[i] = crop of utterance utt[i] at frame off[i], T frames,
 * right-zero-padded when the utterance is shorter (:100-101). */
int dvae_gather_crop(const float* mels, const int* lens, const int* utt, const int* off, float* out, int n,
                     int C, int T, int Lmax, void* stream);

/* ---- inference: mel -> mel conversion plumbing (voice_conversion_mel, variational_base_vae.py:269-298; chunking_mel :335-348)
 * dvae_mel_to_chunks: mel[C,L] -> out[n,C,T], n = L/T + 1, tail zero-padded (an all-zero chunk when L % T == 0).
 * dvae_chunks_to_mel: in[n,C,T] -> out[C, n*T] (torch.cat of the chunks along time), optional clamp to [lo,hi] (:296).
 * dvae_conversion_latents: z_src[k] = [mean_k style_mu(src) | content_mu(src)[k]], z_conv[k] = [mean style_mu(trg) | same] (:281-285).
 * dvae_mul_div: out = a * (b / c)  (spectral detail, :301). */
int dvae_mel_to_chunks(const float* mel, float* out, int C, int L, int T, int n, void* stream);
int dvae_chunks_to_mel(const float* in, float* out, int n, int C, int T, float lo, float hi, int clamp, void* stream);
int dvae_conversion_latents(const float* src_style, const float* src_content, const float* trg_style,
                            float* z_src, float* z_conv, int n, int m, int S, int Cn, void* stream);
int dvae_mul_div(const float* a, const float* b, const float* c, float* out, int64_t n, void* stream);

/* ---- mel front-end: preprocessing/utils.py:68-73 `melspectrogram` = lws STFT (1024/256, "speech" window)
 * -> |.| -> 80-mel projection (librosa.filters.mel) -> dB (:127-129) - ref_level_db -> normalise to [0,1] (:136-137).
 * The two contractions (frames x DFT basis, magnitudes x mel basis) are dvae_gemm_f32 launches; these are the passes around them.
 * dvae_stft_frames: frames[M, fsize] = window * signal zero-padded by `left` samples in front (lws pads fsize-hop, :82-103).
 * dvae_stft_magnitude: reim[rows, 2*nbp] (real block | imaginary block) -> mag[rows, nbp].
 * dvae_mel_db_normalize: mel[M, n_mels] (frame-major) -> out[n_mels, ld_out] at column col0 (the [80, L] layout of the .npy corpus). */
int dvae_stft_frames(const float* wav, int64_t n, const float* window, float* frames, int M, int fsize, int hop,
                     int left, void* stream);
int dvae_stft_magnitude(const float* reim, float* mag, int64_t rows, int nbins_padded, void* stream);
int dvae_mel_db_normalize(const float* mel, float* out, int M, int n_mels, int64_t ld_out, int64_t col0,
                          float min_level, float ref_level_db, float min_level_db, void* stream);

/* ---- opt-in per-family kernel timing with HIP events on the launch stream (bench.py roofline) ----
 * family: 0 = off, 1 = GEMM/conv contraction kernel, 2 = LSTM step kernels.
 * dvae_prof_collect: synchronises the recorded events, returns total ms, launch count and algorithmic FLOPs. */
int dvae_prof_enable(int family);
int dvae_prof_collect(double* total_ms, int64_t* launches, double* flops);
/* the same, split by kernel instantiation (call BEFORE dvae_prof_collect, which resets).  Contraction family: tag =
 * a_kcontig | b_kcontig << 1 | NTW << 2 | BK << 4 | WG << 10 | mode << 13 | tap_mode << 15, i.e. the template arguments
 * of gemm_f32_kernel<A_KC, B_KC, NTW, BK, WG, MODE>; bytes = algorithmic operand bytes (each element once).  Returns the
 * number of distinct tags (<= max_tags written). */
int dvae_prof_collect_tags(unsigned* tags, double* ms, int64_t* launches, double* flops, double* bytes, int max_tags);
#ifdef __cplusplus
}
#endif
#endif
