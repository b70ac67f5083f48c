import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from test_hip_model import make, synthetic_pair, synthetic_eps, rel
from dvae_amd import ops
V = os.environ.get("V", "")
if "nopers" in V: ops.LSTM_PERSISTENT = False
if "noarena" in V:
    ops.SplitKArena.begin = lambda self, dev: (self.reserve(dev), setattr(self, "off", 0))[0]
B, T = 4, 64
a, b = make(B, T, lr=0.0), make(B, T, lr=0.0)
b.enable_graph(True)
for i in range(4):
    x1, x2 = (t.cuda() for t in synthetic_pair(B, T, 100 + i))
    eps = synthetic_eps(B, seed=200 + i)
    a.model.eps_override = eps; b.model.eps_override = eps
    la = a.step(x1, x2, None, train=True); lb = b.step(x1, x2, None, train=True)
if "reduce" in V:
    for w in (a, b):
        for lo, hi in w.optimizer._zero_ranges:
            assert float(w.optimizer.flat_g[lo:hi].abs().max()) == 0.0
x1, x2 = (t.cuda() for t in synthetic_pair(B, T, 150))
eps = synthetic_eps(B, seed=250)
b.model.eps_override = eps
if "nomanual" not in V:
    b.optimizer.zero_grad()
    if "fwdonly" in V:
        with torch.no_grad():
            b.loss_functionGVAE2(x1, x2, *b.model(x1, x2), train=True)
    else:
        b.loss_functionGVAE2(x1, x2, *b.model(x1, x2), train=True)[0].backward()
if "sync" in V: torch.cuda.synchronize()
a.model.eps_override = eps
if "border" in V:
    lb = b.step(x1, x2, None, train=True); la = a.step(x1, x2, None, train=True)
else:
    la, lb = a.step(x1, x2, None, train=True), b.step(x1, x2, None, train=True)
lb2 = b.step(x1, x2, None, train=True)
print(V or "base", "first", " ".join(f"{rel(p,q):.0e}" for p, q in zip(la, lb)), "| second", " ".join(f"{rel(p,q):.0e}" for p, q in zip(la, lb2)))
