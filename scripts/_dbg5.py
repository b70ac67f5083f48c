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
if "atenzero" in V:
    _ops.zeros = lambda shape, dev: torch.zeros(shape, device=dev, dtype=torch.float32)
    from dvae_amd.optim import FlatAdam
    def _zg(self, set_to_none=False):
        for lo, hi in self._zero_ranges: self.flat_g[lo:hi].zero_()
    FlatAdam.zero_grad = _zg
if "nopers" in V: _ops.LSTM_PERSISTENT = False
B, T = 4, 64
pre = [make(B, T, lr=0.0) for _ in range(2)] if "pre" in V else []
for w in pre:
    x1, x2 = (t.cuda() for t in synthetic_pair(B, T, 100)); w.model.eps_override = synthetic_eps(B, seed=200)
    w.step(x1, x2, None, train=True)
bad = 0
for rep in range(int(os.environ.get("REPS", 8))):
    c = make(B, T)
    c.enable_graph(os.environ.get("GRAPH", "1") == "1")
    x1, x2 = (t.cuda() for t in synthetic_pair(B, T, 7))
    hist = [c.step(x1, x2, None, train=True)[0] for _ in range(6)]
    if any(math.isnan(h) for h in hist) or not hist[-1] < hist[0]:
        bad += 1
        print("  ", rep, [f"{h:.1f}" for h in hist])
    del c
print(V or "base", "bad", bad)
