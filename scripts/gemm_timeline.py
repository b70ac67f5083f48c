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
    nwg = 512
elif kind == "lone":      # 256 workgroups: one per CU, no co-resident competitor
    x, w, y = t(R, 1024), t(256, 1024), torch.empty(R, 256, device="cuda")
    fn = lambda: ops.gemm(x, w, y, None, R, 256, 1024, 1024, 1024, 256, True, True)
    nwg = 256
else:
    x, w, y = t(R, 1024), t(1024, 1024), torch.empty(R, 1024, device="cuda")
    fn = lambda: ops.gemm(x, w, y, None, R, 1024, 1024, 1024, 1024, 1024, True, True)
    nwg = 1024
for _ in range(3):
    fn()
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (nwg * 8))()
L.dvae_probe_gemm_timeline.restype = ctypes.c_int
assert L.dvae_probe_gemm_timeline(ctypes.cast(buf, ctypes.c_void_p), nwg * 8) == 0
a = np.array(buf, dtype=np.uint64).reshape(nwg, 8).astype(np.float64)
it = a[:, 1].mean()
print(f"{kind}: {nwg} workgroups, {it:.0f} k-tiles each: {a[:, 0].mean() / it:.0f} cycles per k-tile per wave "
      f"(64 MFMAs x 64 cycles = 4096 per wave; with two workgroups per CU two waves share a SIMD: 8192 per pair)")
