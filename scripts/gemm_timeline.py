"""GPU-box (library built with `csrc/build.sh -DDVAE_GEMM_TS`): cycles per k-tile of wave 0 of every workgroup of ONE
contraction launch.  usage: gemm_timeline.py conv|big|lone"""
import ctypes, os, sys
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
    x.relu_()
    nwg = 512
elif kind == "lone":      # 256 workgroups: one per CU, no co-resident competitor
    x, w, y = t(R, 1024), t(256, 1024), torch.empty(R, 256, device="cuda")
    fn = lambda: ops.gemm(x, w, y, None, R, 256, 1024, 1024, 1024, 256, True, True)
    nwg = 256
else:
    x, w, y = t(R, 1024), t(1024, 1024), torch.empty(R, 1024, device="cuda")
    fn = lambda: ops.gemm(x, w, y, None, R, 1024, 1024, 1024, 1024, 1024, True, True)
    nwg = 1024
for _ in range(int(os.environ.get("REPS", "30"))):     # sustained: the clock settles after a few launches
    fn()
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (nwg * 8))()
L.dvae_probe_gemm_timeline.restype = ctypes.c_int
assert L.dvae_probe_gemm_timeline(ctypes.cast(buf, ctypes.c_void_p), nwg * 8) == 0
a = np.array(buf, dtype=np.uint64).reshape(nwg, 8).astype(np.float64)
it = a[:, 1].mean()
print(f"shader clock during the k loops: {100.0 * a[:, 0].sum() / max(1.0, a[:, 2].sum()):.0f} MHz (s_memtime / s_memrealtime)")
if os.environ.get("TALL") == "1":   # tall kernel: rows = (workgroup, wave); steps 0-17 | 18-35 | 36-47 | barrier
    r = a[a[:, 1] > 0]
    n = r[:, 1].mean()
    print(f"  tall: entry -> loop {r[:, 7].mean():.0f} cycles; loop {r[:, 0].mean():.0f}; epilogue: stores issued after {r[:, 4].mean():.0f}, drained after {r[:, 5].mean():.0f} "
          f"(slowest wave: {r[:, 7].max():.0f} / {r[:, 0].max():.0f} / {r[:, 5].max():.0f})")
    print(f"  tall: per k-tile  steps 0-17 {r[:, 3].mean() / n:.0f}  18-35 {r[:, 4].mean() / n:.0f}  36-47 {r[:, 5].mean() / n:.0f}  barrier {r[:, 6].mean() / n:.0f}  "
          f"total {r[:, 0].mean() / n:.0f} cycles ({len(r)} waves; 48 MFMAs = 1536)")
elif a[:, 3].sum() > 0:   # ping-pong kernel: per-phase cycles (M stage, barrier, C compute, barrier) per k-tile
    for hh in (0, 1):
        r = a[hh::2]
        r = r[r[:, 1] > 0]
        n = r[:, 1].mean()
        print(f"  half {hh}: per k-tile  M {r[:, 3].mean() / n:.0f}  wait {r[:, 4].mean() / n:.0f}  C {r[:, 5].mean() / n:.0f}  wait {r[:, 6].mean() / n:.0f}  "
              f"total {r[:, 0].mean() / n:.0f} cycles ({len(r)} half-workgroups)")
print(f"{kind}: {nwg} workgroups, {it:.0f} k-tiles each: {a[:, 0].mean() / it:.0f} cycles per k-tile per wave "
      f"(64 MFMAs x 64 cycles = 4096 per wave; with two workgroups per CU two waves share a SIMD: 8192 per pair)")
