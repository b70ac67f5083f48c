/*
 * dvae_hip_dev.h — entry points that exist only in the DEVELOPMENT build of the library (csrc/build.sh dev ->
 * libdvae_dev.so, -DDVAE_DEV -DDVAE_PERS_TS): hardware probes and in-kernel timelines used by scripts/.  The product
 * library (libdvae_hip.so) exports none of them and reads no environment variable in its kernel dispatch.
 */
#ifndef DVAE_HIP_DEV_H
#define DVAE_HIP_DEV_H
#include "dvae_hip.h"
#ifdef __cplusplus
extern "C" {
#endif
/* experiments: n back-to-back launches of an empty kernel (launch-floor probe, scripts/launch_floor.py) */
int dvae_probe_launches(int n, int blocks, int threads, int lds_bytes, float* sink, void* stream);
/* experiments: register-only MFMA chains (shape 32 -> 32x32x2 f32, else 16x16x4 f32): the matrix-pipe ceiling of THIS chip */
int dvae_probe_mfma(int blocks, int iters, int shape, float* out, void* stream);
/* bf16 matrix-pipe ceiling with the split-mode MFMA stream on register operands; pattern 0 zeros, 1 random constant,
 * 2 random changing every k-step; out2[0..1] = shader-clock cycles and 100 MHz reference ticks of block 0's loop */
/* do the split arithmetic (VALU) of one wave and the MFMAs of another wave of the same SIMD overlap?  which: 1 MFMA waves,
 * 2 VALU waves, 3 both; out2[0] / out2[1] = cycles of an MFMA / a VALU wave of block 0 */
int dvae_probe_coissue(int blocks, int iters, int which, int prio, float* out, unsigned long long* out2, void* stream);
int dvae_probe_mfma_bf16(int blocks, int iters, int pattern, float* out, unsigned long long* out2, void* stream);
/* stamps of workgroup `bid` of the persistent LSTM launches go to buf[frame][wave 0..7][8] (s_memrealtime, 100 MHz) */
int dvae_lstm_pers_set_ts(void* buf, int bid);
/* fp32x3 forward persistent kernel: register dumps (dbg[frame][workgroup][thread][12 words]: xor-fold of the loaded h fragments,
 * the pre-activations used, the gate sums; dbg_frag[frame][row group][wave][unit*3+plane][lane][4 words]: raw fragments of the
 * workgroups with jb == dbg_jb) and a ring of nslot slots instead of two (scripts/x3_fwd16_diag2.py); NULL / 2 switch it off */
int dvae_lstm_pers_set_dbg(void* dbg, void* dbg_frag, int dbg_jb, int nslot);
int64_t dvae_lstm_pers_ws_bytes_slots(int N, int H, int nslot);
#ifdef __cplusplus
}
#endif
#endif
