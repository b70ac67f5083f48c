"""GPU-box: in-kernel timeline of the H=1024 forward frame kernel (DVAE_LSTM_DBG=8 must be set in the environment)."""
import ctypes
import os
import sys

os.environ.setdefault("DVAE_LSTM_DBG", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import dvae_amd  # noqa
from dvae_amd import ops
from dvae_amd._lib import lib

H, In = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1024, 512)
T, N = 128, 128
P = lambda *s: torch.nn.Parameter(torch.randn(*s, device="cuda") * 0.05)
ps = [P(4 * H, In), P(4 * H, H), P(4 * H), P(4 * H)] + [None] * 4
x = torch.randn(T * N, In, device="cuda")
with torch.no_grad():
    for _ in range(2):
        ops.LstmLayerFn.apply(x, T, N, *ps)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (256 * 8))()
assert lib().dvae_probe_lstm_timeline(ctypes.cast(buf, ctypes.c_void_p), 256 * 8) == 0
ts = np.array(buf, dtype=np.uint64).reshape(256, 8).astype(np.int64)
t0 = ts[:, 0].min()
rel = (ts[:, :7] - t0).astype(np.float64)
names = ["entry", "tile0 staged", "round0 computed", "tile1 staged", "acc parked", "reduced", "done"]
print("s_memtime ticks relative to the earliest workgroup entry (256 workgroups): min / median / max")
for i, n in enumerate(names):
    print(f"  {n:16s} {rel[:, i].min():9.0f} {np.median(rel[:, i]):9.0f} {rel[:, i].max():9.0f}")
wall = ts[:, 7]
print("wall_clock64 spread of 'done' (10 ns ticks):", wall.max() - wall.min())
d = rel[:, 1:] - rel[:, :-1]
print("median per-phase ticks:", [int(np.median(d[:, i])) for i in range(6)])
