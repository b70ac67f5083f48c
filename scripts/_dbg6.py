import os, sys, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from test_hip_model import make, synthetic_pair, synthetic_eps, rel
from dvae_amd import ops as _ops
V = os.environ.get("V", "")
if "nofan" in V:
    import dvae_amd.model.disentangled_vae as _m
    _m.fanout = lambda x, n: (x,) * n
if "nopers" in V: _ops.LSTM_PERSISTENT = False
B, T = 4, 64
a, b = make(B, T, lr=0.0), make(B, T, lr=0.0)
b.enable_graph(True)
for i in range(4):
    x1, x2 = (t.cuda() for t in synthetic_pair(B, T, 100 + i))
    eps = synthetic_eps(B, seed=200 + i)
    a.model.eps_override = eps; b.model.eps_override = eps
    la = a.step(x1, x2, None, train=True); lb = b.step(x1, x2, None, train=True)
for (n1, v1), (n2, v2) in zip(a.model.named_buffers(), b.model.named_buffers()):
    if n1.endswith("num_batches_tracked"):
        assert int(v1) == int(v2) == 8
    else:
        assert float((v1 - v2).abs().max()) <= 1e-4 * max(1.0, float(v1.abs().max())), n1
msg = ""
if "nointer" not in V:
    x1, x2 = (t.cuda() for t in synthetic_pair(B, T, 150))
    eps = synthetic_eps(B, seed=250)
    b.model.eps_override = eps
    b.optimizer.zero_grad()
    b.loss_functionGVAE2(x1, x2, *b.model(x1, x2), train=True)[0].backward()
    a.model.eps_override = eps
    la, lb = a.step(x1, x2, None, train=True), b.step(x1, x2, None, train=True)
    msg = f"inter {max(rel(p, q) for p, q in zip(la, lb)):.0e}"
c = make(B, T)
c.enable_graph(True)
x1, x2 = (t.cuda() for t in synthetic_pair(B, T, 7))
hist = [c.step(x1, x2, None, train=True)[0] for _ in range(5)]
ok = (not any(math.isnan(h) for h in hist)) and hist[-1] < hist[0]
print(V or "base", msg, "OK" if ok else "BAD " + str([f"{h:.1f}" for h in hist]))
