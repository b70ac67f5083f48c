"""GPU box (dev build): dvae_colsum_add on the shapes of the step, against DVAE_COLSUM_ROWS (rows per workgroup)."""
import os as _os
_os.environ.setdefault("DVAE_LIB_PATH", _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
                                                       "disentangle-vae-for-vc_amd", "libdvae_dev.so"))
import sys
sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
import torch
import dvae_amd  # noqa
from dvae_amd import ops
for R, C, dt in ((65536, 512, torch.bfloat16), (16384, 512, torch.float32), (65536, 256, torch.float32),
                 (16384, 4096, torch.float32), (16384, 80, torch.float32), (65536, 512, torch.float32)):
    x = torch.randn(R, C, device="cuda").to(dt)
    o = torch.zeros(C, device="cuda")
    for _ in range(5):
        ops.colsum_add(x, o)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        ops.colsum_add(x, o)
    e1.record()
    torch.cuda.synchronize()
    us = 20.0 * e0.elapsed_time(e1)
    print(f"rows={_os.environ.get('DVAE_COLSUM_ROWS', 'auto'):>5s} [{R} x {C}] {str(dt)[6:]:9s} {us:7.1f} us  {x.numel() * x.element_size() / us / 1e6:5.2f} TB/s")
