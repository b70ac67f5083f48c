"""GPU box: which ATen operators does one eager train step still dispatch on device tensors (fills, adds, copies, cat,
RNG)?  A TorchDispatchMode logs every aten op with its output shape and the innermost frame inside this repo (ops the
autograd engine issues itself — gradient accumulation, zero materialisation — show the frame of the backward call)."""
import os
import sys
import traceback
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.utils._python_dispatch import TorchDispatchMode
import bench
from dvae_amd.data import SyntheticPairs

B, T = int(os.environ.get("B", 64)), int(os.environ.get("T", 128))
dtype = os.environ.get("DVAE_COMPUTE_DTYPE", "fp32x3")
dev = torch.device("cuda", 0)
w = bench.build_trainer(dev, B, T, dtype)
x1, x2, spk = SyntheticPairs(B, T, n_speakers=10, seed=1234, device=dev).batch()
for _ in range(2):
    w.step(x1, x2, spk, train=True)
torch.cuda.synchronize()
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
agg = defaultdict(int)
VIEWS = ("view", "reshape", "as_strided", "detach", "alias", "expand", "permute", "transpose", "t.", "slice", "select",
         "unsqueeze", "squeeze", "empty", "_unsafe_view", "narrow", "unbind", "split", "is_", "size", "stride", "lift")


class Log(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = str(func)
        if not any(v in name for v in VIEWS):
            t = out if isinstance(out, torch.Tensor) else next((a for a in args if isinstance(a, torch.Tensor)), None)
            if t is not None and t.is_cuda:
                fr = [f for f in traceback.extract_stack() if root in f.filename and "aten_sources" not in f.filename]
                where = f"{os.path.relpath(fr[-1].filename, root)}:{fr[-1].lineno}" if fr else "(autograd engine)"
                agg[(name, tuple(t.shape), where)] += 1
        return out


with Log():
    w._eager_train_step(x1, x2) if hasattr(w, "_eager_train_step") else w.step(x1, x2, spk, train=True)
torch.cuda.synchronize()
for (name, shape, where), n in sorted(agg.items(), key=lambda kv: (kv[0][2], kv[0][0])):
    print(f"{n:3d}  {name:34s} {str(shape):22s} {where}")
