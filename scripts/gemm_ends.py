"""GPU box (dev library built with `csrc/build.sh dev -DDVAE_GEMM_TS2`): where a tall-kernel launch spends the time that is
NOT its k loop.  Four s_memrealtime stamps (100 MHz, chip-wide) per wave: entry, loop entered, loop left, stores drained.
usage: gemm_ends.py conv|proj|wgrad|dgrad"""
import ctypes, os, sys
os.environ.setdefault("DVAE_LIB_PATH", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                    "disentangle-vae-for-vc_amd", "libdvae_dev.so"))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import dvae_amd  # noqa
from dvae_amd import ops
from dvae_amd._lib import check, lib, ptr, stream
kind = sys.argv[1] if len(sys.argv) > 1 else "conv"
R, N = 16384, 128
L = lib()
t = lambda *s: torch.randn(*s, device="cuda")
if kind == "conv":
    x, wp, b, y = t(R, 512), t(5, 512, 512), t(512), torch.empty(R, 512, device="cuda")
    fn = lambda: check(L.dvae_conv5_fwd(ptr(x), ptr(wp), ptr(b), ptr(y), R, N, 512, 512, -1, stream()), "")
elif kind == "proj":
    x, w, y = t(R, 1024), t(4096, 1024), torch.empty(R, 4096, device="cuda")
    fn = lambda: ops.gemm(x, w, y, None, R, 4096, 1024, 1024, 1024, 4096, True, True)
elif kind == "wgrad":     # conv weight gradient: 5 taps x 6 k-splits x 8 tiles, atomically accumulated
    dy, x, dw = t(R, 512), t(R, 512), torch.zeros(5, 512, 512, device="cuda")
    fn = lambda: check(L.dvae_conv5_wgrad(ptr(dy), ptr(x), ptr(dw), R, N, 512, 512, 6, -1, stream()), "")
elif kind == "dgrad":
    dy, wpt, dx = t(R, 512), t(5, 512, 512), torch.empty(R, 512, device="cuda")
    fn = lambda: check(L.dvae_conv5_dgrad_t(ptr(dy), ptr(wpt), ptr(dx), R, N, 512, 512, -1, stream()), "")
else:
    raise SystemExit("conv|proj|wgrad|dgrad")
for _ in range(30):
    fn()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    fn()
e1.record()
torch.cuda.synchronize()
print(f"{kind}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per launch (events, back to back)")
buf = (ctypes.c_ulonglong * (1024 * 8))()
L.dvae_probe_gemm_timeline.restype = ctypes.c_int
assert L.dvae_probe_gemm_timeline(ctypes.cast(buf, ctypes.c_void_p), 1024 * 8) == 0
a = np.array(buf, dtype=np.uint64).reshape(1024, 8).astype(np.float64)
a = a[a[:, 4] > 0]
us = lambda v: v / 100.0
t0 = a[:, 0].min()
ent, l0, l1, end = (us(a[:, i] - t0) for i in range(4))
print(f"  {len(a)} waves of the first 256 workgroups (z = 0), {a[:, 4].mean():.0f} k-tiles each")
print(f"  entry after the first wave's entry: mean {ent.mean():.2f} us, max {ent.max():.2f}")
print(f"  prologue (entry -> loop):  mean {(l0 - ent).mean():.2f} us  max {(l0 - ent).max():.2f}")
print(f"  k loop:                    mean {(l1 - l0).mean():.2f} us  min {(l1 - l0).min():.2f}  max {(l1 - l0).max():.2f}")
print(f"  epilogue (loop -> drained): mean {(end - l1).mean():.2f} us  max {(end - l1).max():.2f}")
print(f"  last wave drained at {end.max():.2f} us; mean wave drained at {end.mean():.2f} (tail = {end.max() - end.mean():.2f})")
for x_ in range(8):
    m = a[:, 5] == x_
    if m.any():
        print(f"    XCD {x_}: {int(m.sum())} waves  loop {(l1 - l0)[m].mean():.2f} us  drained at {end[m].mean():.2f} (max {end[m].max():.2f})")
