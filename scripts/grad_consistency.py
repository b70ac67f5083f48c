"""GPU box: are the gradients of the train step the same every time?  lr = 0 and the same input and noise every step:
after n steps Adam's moments are m = g (1 - b1^n) and v = g^2 (1 - b2^n) element by element IF every step produced the
same g, so m^2 / v must equal (1 - b1^n)^2 / (1 - b2^n) everywhere; an element whose gradient was lost, stale or doubled
in even one step stands out (this needs no reference run).  The step's atomically accumulated sums make g noisy at the
1e-6 level; elements that are tiny against their tensor's scale are skipped."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dvae_amd
from dvae_amd import ops
from oracle.fill import fill_state_dict, synthetic_eps, synthetic_pair

B, T, N = int(os.environ.get("B", 4)), int(os.environ.get("T", 64)), int(os.environ.get("N", 40))
ops.set_compute_dtype(os.environ.get("DVAE_COMPUTE_DTYPE", "fp32x3"))


def make():
    w = dvae_amd.ConvolutionalMulVAE("VCTK", T, 80, 32, 0.0, 0.01, 500, False, batch_size=B, speaker_size=4,
                                     device=torch.device("cuda"), latent_dim=32, mse_cof=10, kl_cof=10)
    w.model.load_state_dict(fill_state_dict(w.model.state_dict()))
    w.model.train()
    return w


x1, x2 = (t.cuda() for t in synthetic_pair(B, T, 150))
eps = synthetic_eps(B, seed=250)
for graph in (True, False):
    w = make()
    w.enable_graph(graph)
    w.model.eps_override = eps
    other = make() if os.environ.get("OTHER") else None          # a second (eager) trainer stepping in between
    if other is not None:
        other.model.eps_override = eps
    for i in range(N):
        w.step(x1, x2, None, train=True)
        if other is not None and i % 2:
            other.step(x1, x2, None, train=True)
    o = w.optimizer
    b1, b2 = o.betas
    want = (1 - b1 ** N) ** 2 / (1 - b2 ** N)
    worst, bad, total = 0.0, 0, 0
    worst_name = ""
    for n, p in zip(o.names, o.params):
        if n.endswith(".0.conv.bias") or (n.startswith("dec_modules.") and n.endswith(".0.bias")):
            continue                                            # pre-BatchNorm conv biases: the gradient is round-off
        lo = o.offsets[n]
        m, v = o.exp_avg[lo:lo + p.numel()].double(), o.exp_avg_sq[lo:lo + p.numel()].double()
        g = m.abs() / (1 - b1 ** N)
        keep = g > 1e-2 * g.max()
        if not bool(keep.any()):
            continue
        r = (m[keep] ** 2 / v[keep] / want - 1).abs()
        total += int(keep.sum())
        bad += int((r > 1e-3).sum())
        if float(r.max()) > worst:
            worst, worst_name = float(r.max()), n
    print(f"graph={graph} other={other is not None}: {total} elements checked, {bad} off by > 1e-3, worst {worst:.2e} ({worst_name})",
          flush=True)
