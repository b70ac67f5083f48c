"""ctypes binding of libdvae_hip.so (C ABI declared in include/dvae_hip.h).

The library is the product: if it is missing or a call fails, this module raises —
there is no PyTorch/CPU fallback anywhere in the package.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import torch  # noqa: F401  (loads torch's bundled libamdhip64.so.7 first, so ours binds to the same runtime)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DVAE_LIB_PATH") or os.path.join(_HERE, "libdvae_hip.so")   # env: experimental builds (scripts/)
_lib = None

vp, i32, i64, f32 = C.c_void_p, C.c_int, C.c_int64, C.c_float


class Ranges(C.Structure):
    """dvae_ranges_t"""
    _fields_ = [("lo", i64 * 8), ("hi", i64 * 8), ("n", i32)]


class LstmDir(C.Structure):
    """dvae_lstm_dir_t"""
    _fields_ = [("gates", vp), ("w_hh", vp), ("h_out", vp), ("c_all", vp), ("dh_out", vp),
                ("dgates", vp), ("dc_ws", vp), ("w_packed", vp), ("reverse", i32), ("packed_mode", i32),
                ("step_shift", i32), ("state_bf16", i32), ("pers_ws", vp), ("pers_timeout_us", C.c_uint), ("dbias_ih", vp), ("dbias_hh", vp), ("dbias_part", vp), ("gate_ld", i64)]


class SlabDesc(C.Structure):
    """dvae_slab_desc_t"""
    _fields_ = [("c", vp), ("slab", vp), ("slab_stride", i64), ("n", i64), ("nslab", i32), ("pad_", i32)]


SLAB_FOLD_MAX = 64        # DVAE_SLAB_FOLD_MAX
PERS_BIAS_SLABS = 16      # DVAE_PERS_BIAS_SLABS


class RepackDesc(C.Structure):
    """dvae_repack_desc_t"""
    _fields_ = [("kind", i32), ("d0", i32), ("d1", i32), ("d2", i32), ("src", vp), ("src2", vp), ("dst", vp),
                ("dst2", vp)]


class LossDesc(C.Structure):
    """dvae_loss_desc_t"""
    _fields_ = [(k, vp) for k in ("x1", "x2", "recon1", "recon2", "recon1_hat", "recon2_hat", "q1_mu", "q1_lv", "q2_mu",
                                  "q2_lv", "s_mu", "s_lv")] + \
               [("n", i64), ("nq", i32), ("ns", i32), ("l1_scale", f32), ("kl_scale", f32), ("style_scale", f32),
                ("mse_cof", f32), ("kl_cof", f32)]


REPACK_CONV_T, REPACK_LSTM_PACK, REPACK_TRANSPOSE, REPACK_ADD2, REPACK_CAST_BF16, REPACK_COPY_F32 = 0, 1, 2, 3, 4, 5

# DVAE_MODE_* of include/dvae_hip.h
MODE_F32, MODE_BF16, MODE_F32X3 = 0, 1, 2
COMPUTE_MODES = {"fp32": MODE_F32, "f32": MODE_F32, "float32": MODE_F32, "bf16": MODE_BF16, "bfloat16": MODE_BF16,
                 "fp32x3": MODE_F32X3, "f32x3": MODE_F32X3}

DEFAULT_COMPUTE_DTYPE = "fp32x3"
ABI_VERSION = 307     # DVAE_ABI_VERSION of include/dvae_hip.h

# name -> (restype, argtypes); mirrors include/dvae_hip.h one to one
SIGNATURES = {
    "dvae_version": (i32, []),
    "dvae_last_hip_error": (i32, []),
    "dvae_gemm_f32": (i32, [vp, vp, vp, vp, i32, i32, i32, i64, i64, i64, i32, i32, i32, i32, i32, i32, vp]),
    "dvae_conv5_fwd": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, i32, vp]),
    "dvae_conv5_fwd_stats": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp, vp]),
    "dvae_bn_stats_finalize": (i32, [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, f32, f32, vp]),
    "dvae_conv5_wgrad": (i32, [vp, vp, vp, i32, i32, i32, i32, i32, i32, vp]),
    "dvae_conv_pack_w": (i32, [vp, vp, i32, i32, vp]),
    "dvae_conv_pack_wt": (i32, [vp, vp, i32, i32, vp]),
    "dvae_conv5_dgrad_t": (i32, [vp, vp, vp, i32, i32, i32, i32, i32, vp]),
    "dvae_conv_unpack_add_w": (i32, [vp, vp, i32, i32, vp]),
    "dvae_bn_ws_bytes": (i64, [i32, i32, i32]),
    "dvae_bn_stats_fwd": (i32, [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, f32, f32, vp]),
    "dvae_bn_apply_fwd": (i32, [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp]),
    "dvae_bn_bwd": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp]),
    "dvae_bn_bwd_from_y": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp]),
    "dvae_lstm_pack_w": (i32, [vp, vp, vp, i32, vp]),
    "dvae_lstm_pack_w_bf16": (i32, [vp, vp, vp, i32, vp]),
    "dvae_lstm_pack_w_x3": (i32, [vp, vp, vp, i32, vp]),
    "dvae_repack_all": (i32, [C.POINTER(RepackDesc), i32, vp]),
    "dvae_lstm_pers_supported": (i32, [i32, i32, i32, i32]),
    "dvae_lstm_pers_ws_bytes": (i64, [i32, i32]),
    "dvae_lstm_pers_check": (i32, [vp, C.POINTER(i32), vp]),
    "dvae_lstm_pers_selftest": (i32, [C.POINTER(LstmDir), i32, i32, i32, i64, i32, vp]),
    "dvae_lstm_seq_fwd": (i32, [C.POINTER(LstmDir), i32, i32, i32, i32, i64, vp]),
    "dvae_lstm_seq_bwd": (i32, [C.POINTER(LstmDir), i32, i32, i32, i32, i64, vp]),
    "dvae_lstm_seq_fwd_range": (i32, [C.POINTER(LstmDir), i32, i32, i32, i32, i64, i32, i32, vp]),
    "dvae_lstm_seq_bwd_range": (i32, [C.POINTER(LstmDir), i32, i32, i32, i32, i64, i32, i32, vp]),
    "dvae_latent_fwd": (i32, [vp] * 9 + [i32, i32, i32, vp]),
    "dvae_latent_bwd": (i32, [vp] * 11 + [i32, i32, i32, vp]),
    "dvae_kl_fwd": (i32, [vp, vp, vp, i64, f32, vp]),
    "dvae_kl_bwd": (i32, [vp, vp, vp, vp, vp, i64, f32, vp]),
    "dvae_l1_ws_bytes": (i64, [i64]),
    "dvae_l1_sum_fwd": (i32, [vp, vp, vp, vp, i64, f32, vp]),
    "dvae_l1_sum_bwd": (i32, [vp, vp, vp, vp, i64, f32, vp]),
    "dvae_loss_ws_bytes": (i64, [i64]),
    "dvae_loss_fwd": (i32, [C.POINTER(LossDesc), vp, vp, vp]),
    "dvae_loss_bwd": (i32, [C.POINTER(LossDesc), vp] + [vp] * 10 + [vp]),
    "dvae_gemm_f32_batched": (i32, [vp, vp, vp, i32, i32, i32, i32, i64, i64, i64, i32, i32, i32, i32, i32, vp]),
    "dvae_gemm_f32_slabs": (i32, [vp, vp, vp, vp, i64, i32, vp, i32, i32, i32, i64, i64, i64, i32, i32, i32, i32, i32, vp]),
    "dvae_gemm_f32_batched_slabs": (i32, [vp, vp, vp, i32, vp, i64, i32, i32, i32, i32, i64, i64, i64, i32, i32, i32, i32,
                                          i32, vp]),
    "dvae_conv5_fwd_slabs": (i32, [vp, vp, vp, vp, vp, i64, i32, i32, i32, i32, i32, i32, vp]),
    "dvae_conv5_dgrad_t_slabs": (i32, [vp, vp, vp, vp, i64, i32, i32, i32, i32, i32, i32, vp]),
    "dvae_conv5_wgrad_slabs": (i32, [vp, vp, vp, vp, i64, i32, i32, i32, i32, i32, i32, i32, i32, vp]),
    "dvae_slab_sum": (i32, [vp, vp, i64, i32, i64, i32, i32, vp]),
    "dvae_slab_fold": (i32, [C.POINTER(SlabDesc), i32, vp]),
    "dvae_colsum_ws_bytes": (i64, [i32, i32]),
    "dvae_colsum_add_ws": (i32, [vp, vp, vp, i32, i32, i64, i32, vp, vp]),
    "dvae_adam_flat": (i32, [vp, vp, vp, vp, i64, f32, f32, f32, f32, f32, i32, vp]),
    "dvae_adam_flat_dev": (i32, [vp, vp, vp, vp, i64, f32, f32, f32, vp, vp, vp, i32, vp]),
    "dvae_lstm_pers_err_word": (vp, [vp]),
    "dvae_sum_f32": (i32, [vp, vp, vp, vp, i64, vp]),
    "dvae_zero_f32": (i32, [vp, i64, vp]),
    "dvae_mel_to_frames": (i32, [vp, vp, vp, i32, i32, i32, i32, vp]),
    "dvae_frames_to_mel": (i32, [vp, vp, i32, i32, i32, vp]),
    "dvae_permute_102": (i32, [vp, vp, i32, i32, i32, vp]),
    "dvae_colsum_add": (i32, [vp, vp, vp, i32, i32, i64, i32, vp]),
    "dvae_transpose": (i32, [vp, vp, i32, i32, vp]),
    "dvae_act_fwd": (i32, [vp, i64, i32, vp]),
    "dvae_act_bwd": (i32, [vp, vp, vp, i64, i32, vp]),
    "dvae_gather_crop": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, i32, vp]),
    "dvae_mel_to_chunks": (i32, [vp, vp, i32, i32, i32, i32, vp]),
    "dvae_chunks_to_mel": (i32, [vp, vp, i32, i32, i32, f32, f32, i32, vp]),
    "dvae_conversion_latents": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, i32, vp]),
    "dvae_mul_div": (i32, [vp, vp, vp, vp, i64, vp]),
    "dvae_set_compute_mode": (i32, [i32]),
    "dvae_get_compute_mode": (i32, []),
    "dvae_set_deterministic": (i32, [i32]),
    "dvae_get_deterministic": (i32, []),
    "dvae_stft_frames": (i32, [vp, i64, vp, vp, i32, i32, i32, i32, vp]),
    "dvae_stft_magnitude": (i32, [vp, vp, i64, i32, vp]),
    "dvae_mel_db_normalize": (i32, [vp, vp, i32, i32, i64, i64, f32, f32, f32, vp]),
    "dvae_prof_enable": (i32, [i32]),
    "dvae_prof_collect": (i32, [C.POINTER(C.c_double), C.POINTER(i64), C.POINTER(C.c_double)]),
    "dvae_prof_collect_tags": (i32, [C.POINTER(C.c_uint), C.POINTER(C.c_double), C.POINTER(i64), C.POINTER(C.c_double),
                                     C.POINTER(C.c_double), i32]),
}


# include/dvae_hip_dev.h: only in the development build (csrc/build.sh dev -> libdvae_dev.so; DVAE_LIB_PATH selects it)
DEV_SIGNATURES = {
    "dvae_probe_launches": (i32, [i32, i32, i32, i32, vp, vp]),
    "dvae_probe_mfma": (i32, [i32, i32, i32, vp, vp]),
    "dvae_probe_mfma_bf16": (i32, [i32, i32, i32, vp, vp, vp]),
    "dvae_probe_coissue": (i32, [i32, i32, i32, i32, vp, vp, vp]),
    "dvae_lstm_pers_set_ts": (i32, [vp, i32]),
    "dvae_lstm_pers_set_dbg": (i32, [vp, vp, i32, i32]),
    "dvae_lstm_pers_ws_bytes_slots": (i64, [i32, i32, i32]),
}


def build(force: bool = False) -> str:
    """Compile the HIP sources for gfx950 into libdvae_hip.so (in-tree)."""
    srcs = [os.path.join(_HERE, "csrc", f) for f in os.listdir(os.path.join(_HERE, "csrc"))
            if f.endswith((".hip", ".h"))]
    if not force and os.path.exists(LIB_PATH):
        newest = max(os.path.getmtime(s) for s in srcs)
        if os.path.getmtime(LIB_PATH) >= newest:
            return LIB_PATH
    # force: every object is recompiled (FORCE=1), not just relinked from objects a previous build left behind
    env = dict(os.environ, FORCE="1") if force else None
    subprocess.check_call(["bash", os.path.join(_HERE, "csrc", "build.sh")], env=env)
    return LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: the HIP extension is the product path and there is no fallback. "
                "Build it with `python -c 'import __graft_entry__ as g; g.build()'` (needs hipcc).")
        h = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(h, name)  # AttributeError if the .so lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        for name, (res, args) in DEV_SIGNATURES.items():
            if hasattr(h, name):
                getattr(h, name).restype, getattr(h, name).argtypes = res, args
        got = h.dvae_version()
        if got != ABI_VERSION:
            raise RuntimeError(f"{LIB_PATH} reports ABI {got}, this binding is for {ABI_VERSION} (include/dvae_hip.h): "
                               "stale library — rebuild it with `python -c 'import __graft_entry__ as g; g.build()'`")
        _lib = h
        # process default of the contraction arithmetic (ops.set_compute_dtype changes it): fp32 results on the bf16
        # matrix pipe unless the environment says otherwise
        mode = os.environ.get("DVAE_COMPUTE_DTYPE", DEFAULT_COMPUTE_DTYPE).lower()
        if mode in COMPUTE_MODES:
            h.dvae_set_compute_mode(COMPUTE_MODES[mode])
        elif mode:
            raise RuntimeError(f"DVAE_COMPUTE_DTYPE={mode!r}: expected one of {sorted(COMPUTE_MODES)}")
    return _lib


class DvaeHipError(RuntimeError):
    pass


def check(rc: int, what: str):
    if rc != 0:
        err = lib().dvae_last_hip_error() if rc == -2 else 0
        raise DvaeHipError(f"{what} failed: rc={rc} hipError={err}")


def ptr(t):
    """Device pointer of a tensor (None -> NULL).  Tensors must be contiguous fp32 unless noted."""
    if t is None:
        return None
    return t.data_ptr()


def stream() -> int:
    return torch.cuda.current_stream().cuda_stream
