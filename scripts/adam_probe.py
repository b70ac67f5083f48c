import os, sys
sys.path.insert(0, "/root/repo")
import torch
import dvae_amd
from dvae_amd.optim import FlatAdam
n = 94930304
p = torch.nn.Parameter(torch.randn(n, device="cuda"))
opt = FlatAdam([("p", p)], lr=1e-4)
opt.guard_device_errors = False
p.grad.normal_()
for _ in range(5): opt.step()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): opt.step()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 50
print(os.environ.get("DVAE_LIB_PATH", "product")[-12:], f"{ms*1e3:.1f} us  {n*32/ms/1e9:.2f} TB/s (8 x 4 B per element with the clear)")
