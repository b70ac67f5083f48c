"""GPU-box: the 512->512 conv forward launch, sustained, over ONE operand set (everything stays in the 256 MB Infinity
Cache) vs rotating over S sets (S x 72 MB: operands come from HBM, as inside the train step).  usage: conv_rotating.py [S]"""
import os as _os
# needs the DEVELOPMENT build of the library (csrc/build.sh dev): probes / environment knobs / timelines are not in the product
_os.environ.setdefault("DVAE_LIB_PATH", _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
                                                       "disentangle-vae-for-vc_amd", "libdvae_dev.so"))
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dvae_amd  # noqa
from dvae_amd._lib import check, lib, ptr, stream
R, N = 16384, 128
L = lib()
S = int(sys.argv[1]) if len(sys.argv) > 1 else 8
reps = 2000
sets = [(torch.randn(R, 512, device="cuda"), torch.randn(5, 512, 512, device="cuda"), torch.randn(512, device="cuda"),
         torch.empty(R, 512, device="cuda")) for _ in range(S)]
fl = 2.0 * R * 512 * 512 * 5
for nset in (1, S):
    def run(n):
        for i in range(n):
            x, wp, b, y = sets[i % nset]
            check(L.dvae_conv5_fwd(ptr(x), ptr(wp), ptr(b), ptr(y), R, N, 512, 512, -1, stream()), "")
    run(200)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    run(reps)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"{nset} operand set(s): {ms * 1e3:.1f} us  {fl / ms / 1e9:.1f} TF/s")
