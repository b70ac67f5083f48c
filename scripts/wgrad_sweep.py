"""GPU-box microbench: conv wgrad (tap mode 2) time against the split-K count."""
import os as _os
# needs the DEVELOPMENT build of the library (csrc/build.sh dev): probes / environment knobs / timelines are not in the product
_os.environ.setdefault("DVAE_LIB_PATH", _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
                                                       "disentangle-vae-for-vc_amd", "libdvae_dev.so"))
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dvae_amd  # noqa: F401
from dvae_amd._lib import check, lib, ptr, stream

R, N = 16384, 128
cin, cout = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (512, 512)
L = lib()
x, y = torch.randn(R, cin, device="cuda"), torch.randn(R, cout, device="cuda")
dwp = torch.zeros(5, cout, cin, device="cuda")
fl = 2.0 * R * cin * cout * 5
for sk in (1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 16, 19, 25, 32):
    f = lambda: check(L.dvae_conv5_wgrad(ptr(y), ptr(x), ptr(dwp), R, N, cin, cout, sk, -1, stream()), "")
    for _ in range(100):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(300):
        f()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 300
    print(f"sk={sk:3d} {ms * 1e3:8.1f} us {fl / ms / 1e9:7.1f} TF/s", flush=True)
