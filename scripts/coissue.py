"""GPU box: do VALU work (the split arithmetic) and MFMAs of two different waves of one SIMD overlap?"""
import os as _os
# needs the DEVELOPMENT build of the library (csrc/build.sh dev): probes / environment knobs / timelines are not in the product
_os.environ.setdefault("DVAE_LIB_PATH", _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
                                                       "disentangle-vae-for-vc_amd", "libdvae_dev.so"))
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dvae_amd  # noqa
from dvae_amd._lib import lib, check, stream
L = lib()
out = torch.zeros(4, device="cuda")
iters = 4000
for prio in (0, 1):
    for which, nm in ((1, "MFMA waves only"), (2, "VALU waves only"), (3, "both")):
        o2 = torch.zeros(4, device="cuda", dtype=torch.int64)
        check(L.dvae_probe_coissue(256, 200, which, prio, out.data_ptr(), o2.data_ptr(), stream()), "p"); torch.cuda.synchronize()
        o2.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); check(L.dvae_probe_coissue(256, iters, which, prio, out.data_ptr(), o2.data_ptr(), stream()), "p"); e1.record()
        torch.cuda.synchronize()
        cm, cv = [int(v) for v in o2.tolist()[:2]]
        print(f"prio {prio} {nm:16s}: {e0.elapsed_time(e1):7.3f} ms   MFMA wave {cm / iters:7.0f} cycles / 24 MFMAs (768 = pipe limit)   "
              f"VALU wave {cv / iters:7.0f} cycles / iteration of 4 pieces")
