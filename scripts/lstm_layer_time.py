"""GPU-box: sustained time of one LSTM layer's recurrence launches, forward and backward (T=128, N=128). usage: H In"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dvae_amd  # noqa
from dvae_amd import ops
H, In = int(sys.argv[1]), int(sys.argv[2])
T, N = 128, 128
P = lambda *s: torch.nn.Parameter(torch.randn(*s, device="cuda") * 0.05)
ps = [P(4 * H, In), P(4 * H, H), P(4 * H), P(4 * H)]
for p in ps:
    p.grad = torch.zeros_like(p)
x = torch.randn(T * N, In, device="cuda", requires_grad=True)
gh = torch.randn(T * N, H, device="cuda")
def run():
    h = ops.LstmLayerFn.apply(x, T, N, *ps, None, None, None, None)
    h.backward(gh)
for _ in range(5):
    run()
torch.cuda.synchronize()
ops.prof_enable(2)
for _ in range(10):
    run()
torch.cuda.synchronize()
ms, n, fl = ops.prof_collect(); ops.prof_enable(0)
print(f"H={H}: recurrence fwd+bwd {ms / 10 / (2 * T) * 1e3:.2f} us per frame-launch (avg of fwd and bwd), {fl / ms / 1e9:.1f} TF/s  MT5={os.environ.get('DVAE_LSTM_MT5', 'auto')}")
