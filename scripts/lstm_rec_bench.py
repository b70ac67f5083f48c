"""GPU-box: the LSTM recurrence launches alone (no projections, no weight gradients) through the C ABI.
usage: lstm_rec_bench.py H [stack=0|1] [reps]   (N = 128 segments, T = 128 frames, fp32)
Prints us per layer-frame for forward and backward.  DVAE_LIB_PATH selects an experimental build of the library."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dvae_amd  # noqa: F401
from dvae_amd import _lib
from dvae_amd._lib import check, lib, ptr, stream
from dvae_amd.derived import lstm_local, lstm_pack_modes

H = int(sys.argv[1])
stack = int(sys.argv[2]) if len(sys.argv) > 2 else 0
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
BF = int(os.environ.get("LSTM_MODE", "0"))        # DVAE_MODE_*: 0 fp32 MFMA, 1 bf16, 2 fp32x3 (three bf16 planes)
S16 = int(os.environ.get("LSTM_S16", "0")) if BF == 1 else 0   # bf16 mode: h / dgates stored as bf16
sdt = torch.bfloat16 if S16 else torch.float32
T, N = int(os.environ.get("LSTM_T", "128")), int(os.environ.get("LSTM_N", "128"))
PERS = int(os.environ.get("LSTM_PERS", "0"))      # 1: the W_hh-resident persistent launch (one per sequence; unstacked only)
L = lib()
nl = 2 if stack else 1
dev = "cuda"
f = dict(device=dev, dtype=torch.float32)
layers = []
for l in range(nl):
    w_ih, w_hh = torch.randn(4 * H, 64, **f) * 0.05, torch.randn(4 * H, H, **f) * (1.0 / H ** 0.5)
    b = torch.zeros(4 * H, **f)
    der = lstm_local(w_ih, w_hh, b, b, BF)
    MF, MB = lstm_pack_modes(BF, H)
    layers.append(dict(w_hh=w_hh, der=der, gates0=torch.randn(T * N, 4 * H, **f) * 0.5,
                       gates=torch.empty(T * N, 4 * H, **f), h=torch.empty(T * N, H, device=dev, dtype=sdt),
                       c=torch.empty(T * N, H, **f), dh=torch.randn(T * N, H, **f) * 0.1,
                       dg=torch.empty(T * N, 4 * H, device=dev, dtype=sdt), dc=torch.empty(N, H, **f)))


def dirs(bwd):
    d = (_lib.LstmDir * nl)()
    order = list(reversed(range(nl))) if bwd else list(range(nl))
    for i, l in enumerate(order):
        y = layers[l]
        d[i].gates, d[i].c_all = ptr(y["gates"]), ptr(y["c"])
        d[i].w_hh = ptr(y["der"].w_hh_t if bwd else y["w_hh"])
        d[i].w_packed = ptr(y["der"].pack_b if bwd else y["der"].pack_f)
        d[i].h_out, d[i].dh_out, d[i].dgates, d[i].dc_ws = ptr(y["h"]), ptr(y["dh"]), ptr(y["dg"]), ptr(y["dc"])
        d[i].reverse, d[i].packed_mode, d[i].step_shift = 0, (MB if bwd else MF), (T // 2 if (stack and i == 1) else 0)
        d[i].state_bf16 = S16
        if PERS and not stack:
            from dvae_amd import ops
            d[i].pers_ws = ptr(ops.lstm_pers_workspace(dev))
    return d


def fwd():
    for y in layers:
        y["gates"].copy_(y["gates0"])
    d = dirs(False)
    if stack:
        check(L.dvae_lstm_seq_fwd_range(d, 2, T, N, H, H, 0, T + T // 2, stream()), "fwd")
    else:
        check(L.dvae_lstm_seq_fwd(d, 1, T, N, H, H, stream()), "fwd")


def bwd():
    d = dirs(True)
    if stack:
        check(L.dvae_lstm_seq_bwd_range(d, 2, T, N, H, H, 0, T + T // 2, stream()), "bwd")
    else:
        check(L.dvae_lstm_seq_bwd(d, 1, T, N, H, H, stream()), "bwd")


def timeit(fn, copy_ms=0.0):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps - copy_ms


def copy_only():
    for y in layers:
        y["gates"].copy_(y["gates0"])


cp = timeit(copy_only)
tf = timeit(fwd, cp)
tb = timeit(bwd)
if PERS:
    from dvae_amd import ops
    ops.lstm_pers_check()
per = T * nl
print(f"H={H} N={N} T={T} stack={stack} mode={BF} s16={S16} pers={PERS}: fwd {1e3 * tf / per:.2f} us/layer-frame ({tf:.3f} ms), bwd {1e3 * tb / per:.2f} us/layer-frame "
      f"({tb:.3f} ms)  [lib {os.path.basename(_lib.LIB_PATH)}]", flush=True)
assert torch.isfinite(layers[-1]["h"].float()).all() and torch.isfinite(layers[0]["dg"].float()).all()
