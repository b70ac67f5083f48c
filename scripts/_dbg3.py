import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from test_hip_model import make, synthetic_pair, synthetic_eps, rel
from dvae_amd import ops as _ops
from dvae_amd.optim import FlatAdam
V = os.environ.get("V", "")
_ops.SplitKArena.take = lambda self, shape, dev: torch.zeros(shape, device=dev, dtype=torch.float32)   # arena off
_st = FlatAdam.step
def _step(self, grad_scale=1.0, zero_after=False, clear_extra=None):
    return _st(self, grad_scale, zero_after and getattr(self, "_za", True), None)
FlatAdam.step = _step
B, T = 4, 64
g1, g2, e3 = make(B, T, lr=0.0), make(B, T, lr=0.0), make(B, T, lr=0.0)
g2.optimizer._za = False
g1.enable_graph(True); g2.enable_graph(True)
junk = []
for i in range(40):
    x1, x2 = (t.cuda() for t in synthetic_pair(B, T, 100 + i % 3))
    eps = synthetic_eps(B, seed=200 + i % 3)
    for w in (g1, g2, e3): w.model.eps_override = eps
    if i % 4 == 1:
        for (n1, v1), (n2, v2) in zip(g1.model.named_buffers(), g2.model.named_buffers()):
            junk.append(float((v1.float() - v2.float()).abs().max()))
    if i % 5 == 2:
        e3.optimizer.zero_grad()
        e3.loss_functionGVAE2(x1, x2, *e3.model(x1, x2), train=True)[0].backward()
    order = (e3, g1, g2) if i % 2 else (g2, e3, g1)
    for w in order: w.step(x1, x2, None, train=True)
    if i < 4:
        m1, m2, m3 = g1.optimizer.exp_avg, g2.optimizer.exp_avg, e3.optimizer.exp_avg
        print(i, "norms", float(m1.norm()), float(m2.norm()), float(m3.norm()), "t", g1.optimizer.t, g2.optimizer.t, e3.optimizer.t)
o1, o2, o3 = g1.optimizer, g2.optimizer, e3.optimizer
rows = []
for n, p in zip(o2.names, o2.params):
    lo = o2.offsets[n]; hi = lo + p.numel()
    a, b, c = o1.exp_avg[lo:hi], o2.exp_avg[lo:hi], o3.exp_avg[lo:hi]
    rows.append((float((b - c).norm()) / max(float(c.norm()), 1e-30), float((a - c).norm()) / max(float(c.norm()), 1e-30), n))
for r in sorted(rows, reverse=True)[:12]:
    print(f"g2-e3 {r[0]:.2e}   g1-e3 {r[1]:.2e}   {r[2]}")
