import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from test_hip_bf16 import _make, _dist
from oracle.bf16_ref import RefDVAEBf16
from oracle.dvae_ref import RefDVAE
from oracle.fill import fill_state_dict, synthetic_eps, synthetic_pair
from oracle.dvae_ref import loss_gvae2
from dvae_amd import ops
B, T = 2, 256
x1, x2 = synthetic_pair(B, T, 21)
eps = synthetic_eps(B, seed=22)
grads = {}
for name, cls in (("bf16", RefDVAEBf16), ("fp32", RefDVAE)):
    m = cls(4, 32, T)
    m.load_state_dict(fill_state_dict(m.state_dict()))
    m.train()
    loss_gvae2(x1, x2, m(x1, x2, eps), B)[0].backward()
    grads[name] = {k: p.grad for k, p in m.named_parameters()}
res = []
with ops.compute_dtype("bf16"):
    for rep in range(int(os.environ.get("REPS", 6))):
        w = _make(B, T)
        w.model.eps_override = eps
        w.optimizer.zero_grad()
        outs = w.model(x1.cuda(), x2.cuda())
        w.loss_functionGVAE2(x1.cuda(), x2.cuda(), *outs, train=True)[0].backward()
        row = {}
        for k, p in w.model.named_parameters():
            if k.startswith(("style.", "content.", "enc_lstm.", "enc_linear.")):
                g16, g32, gh = grads["bf16"][k], grads["fp32"][k], w.model.reference_layout(k, p.grad).cpu()
                row[k] = _dist(gh, g16) / max(_dist(g16, g32), 1e-9)
        res.append(row)
for k in res[0]:
    print(f"{k:36s}", " ".join(f"{r[k]:.2f}" for r in res))
