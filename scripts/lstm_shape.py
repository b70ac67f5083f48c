"""GPU-box: time one LSTM layer fwd+bwd sequence (T=128, N=128). usage: lstm_shape.py H In [bidir]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dvae_amd  # noqa
from dvae_amd import ops
H, In = int(sys.argv[1]), int(sys.argv[2])
bidir = len(sys.argv) > 3
T, N = 128, 128
P = lambda *s: torch.nn.Parameter(torch.randn(*s, device="cuda") * 0.05)
ps = [P(4 * H, In), P(4 * H, H), P(4 * H), P(4 * H)]
ps += [P(4 * H, In), P(4 * H, H), P(4 * H), P(4 * H)] if bidir else [None] * 4
x = torch.randn(T * N, In, device="cuda", requires_grad=True)
gy = torch.randn(T * N, (2 if bidir else 1) * H, device="cuda")
def run():
    h = ops.LstmLayerFn.apply(x, T, N, *ps)
    h.backward(gy)
run(); torch.cuda.synchronize()
ops.prof_enable(2)
for _ in range(3): run()
torch.cuda.synchronize()
ms, n, fl = ops.prof_collect(); ops.prof_enable(0)
print(f"H={H} In={In} bidir={bidir}: recurrence fwd+bwd {ms/3:.3f} ms per layer-pass pair, {ms/3/(2*T)*1e3:.2f} us per frame-launch, {fl/ms/1e9:.1f} TF/s")
