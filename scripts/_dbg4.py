import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from test_hip_model import make, synthetic_pair, synthetic_eps, rel
B, T = 4, 64
NT = int(os.environ.get("NT", 2))
g, e = make(B, T, lr=0.0), make(B, T, lr=0.0)
g.enable_graph(True)
extra = [make(B, T, lr=0.0) for _ in range(NT - 2)]
for w in extra: w.enable_graph(True)
junk = []
for i in range(16):
    x1, x2 = (t.cuda() for t in synthetic_pair(B, T, 100 + i % 3))
    eps = synthetic_eps(B, seed=200 + i % 3)
    for w in [g, e] + extra: w.model.eps_override = eps
    if i % 4 == 1 and "junk" in os.environ.get("V", ""):
        for (n1, v1), (n2, v2) in zip(g.model.named_buffers(), e.model.named_buffers()):
            junk.append(float((v1.float() - v2.float()).abs().max()))
    order = [e, g] + extra if i % 2 else extra + [g, e]
    for w in order: w.step(x1, x2, None, train=True)
o1, o2 = g.optimizer, e.optimizer
rows = []
for n, p in zip(o2.names, o2.params):
    lo = o2.offsets[n]; hi = lo + p.numel()
    a, b = o1.exp_avg[lo:hi], o2.exp_avg[lo:hi]
    rows.append((float((a - b).norm()) / max(float(b.norm()), 1e-30), n))
rows = [r for r in rows if not (r[1].endswith(".0.conv.bias") or (r[1].startswith("dec_modules.") and r[1].endswith(".0.bias")))]
print("graph vs eager exp_avg, worst params:", [(f"{r[0]:.1e}", r[1]) for r in sorted(rows, reverse=True)[:4]])
