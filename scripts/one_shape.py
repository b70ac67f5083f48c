"""GPU-box: run ONE contraction shape repeatedly (for rocprofv3 --pmc runs). usage: one_shape.py conv|wgrad|dgrad|gemm [reps]"""
import os as _os
# needs the DEVELOPMENT build of the library (csrc/build.sh dev): probes / environment knobs / timelines are not in the product
_os.environ.setdefault("DVAE_LIB_PATH", _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
                                                       "disentangle-vae-for-vc_amd", "libdvae_dev.so"))
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dvae_amd  # noqa
from dvae_amd import ops
from dvae_amd._lib import check, lib, ptr, stream
kind = sys.argv[1] if len(sys.argv) > 1 else "conv"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2000   # sustained; 20-launch bursts read ~13 % low
R, N = 16384, 128
L = lib()
t = lambda *s: torch.randn(*s, device="cuda")
if kind == "conv":
    x, wp, b, y = t(R, 512), t(5, 512, 512), t(512), torch.empty(R, 512, device="cuda")
    fn = lambda: check(L.dvae_conv5_fwd(ptr(x), ptr(wp), ptr(b), ptr(y), R, N, 512, 512, -1, stream()), "")
    fl = 2.0 * R * 512 * 512 * 5
elif kind == "wgrad":     # conv weight gradient 512 -> 512: 8 tiles x 5 taps x 6 k-splits = 240 workgroups, atomic epilogue
    dy, x, dw = t(R, 512), t(R, 512), torch.zeros(5, 512, 512, device="cuda")
    fn = lambda: check(L.dvae_conv5_wgrad(ptr(dy), ptr(x), ptr(dw), R, N, 512, 512, 6, -1, stream()), "")
    fl = 2.0 * R * 512 * 512 * 5
elif kind == "dgrad":
    dy, wpt, dx = t(R, 512), t(5, 512, 512), torch.empty(R, 512, device="cuda")
    fn = lambda: check(L.dvae_conv5_dgrad_t(ptr(dy), ptr(wpt), ptr(dx), R, N, 512, 512, -1, stream()), "")
    fl = 2.0 * R * 512 * 512 * 5
elif kind == "gemm2560":
    x, w, y = t(R, 2560), t(512, 2560), torch.empty(R, 512, device="cuda")
    fn = lambda: ops.gemm(x, w, y, None, R, 512, 2560, 2560, 2560, 512, True, True)
    fl = 2.0 * R * 512 * 2560
else:
    x, w, y = t(R, 1024), t(4096, 1024), torch.empty(R, 4096, device="cuda")
    fn = lambda: ops.gemm(x, w, y, None, R, 4096, 1024, 1024, 1024, 4096, True, True)
    fl = 2.0 * R * 4096 * 1024
for _ in range(50): fn()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps): fn()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
print(f"{kind}: {ms*1e3:.1f} us {fl/ms/1e9:.1f} TF/s")
