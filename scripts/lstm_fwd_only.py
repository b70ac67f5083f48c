"""GPU-box: time only the forward recurrence launches of one LSTM layer (T=128, N=128). usage: H In"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dvae_amd  # noqa
from dvae_amd import ops
H, In = int(sys.argv[1]), int(sys.argv[2])
T, N = 128, 128
P = lambda *s: torch.nn.Parameter(torch.randn(*s, device="cuda") * 0.05)
ps = [P(4 * H, In), P(4 * H, H), P(4 * H), P(4 * H)] + [None] * 4
x = torch.randn(T * N, In, device="cuda")
with torch.no_grad():
    ops.LstmLayerFn.apply(x, T, N, *ps); torch.cuda.synchronize()
    ops.prof_enable(2)
    for _ in range(3): ops.LstmLayerFn.apply(x, T, N, *ps)
    torch.cuda.synchronize()
    ms, n, fl = ops.prof_collect(); ops.prof_enable(0)
print(f"H={H}: fwd recurrence {ms/3/T*1e3:.2f} us per frame-launch ({fl/ms/1e9:.1f} TF/s) DBG={os.environ.get('DVAE_LSTM_DBG','0')}")
