"""GPU box: an eager trainer and a graph-replayed trainer (same weights, lr = 0, same input and noise) stepping alternately,
as tests/test_hip_model.py::test_graph_replay_matches_eager does.  Per step the gradient of each is recovered from Adam's
first moment (g = (m_t - b1 m_{t-1}) / (1 - b1)) and compared parameter group by parameter group; prints the outlier
steps.  PERS=0 switches the persistent LSTM launches off.  (This is the script that found the flag-clear hazard of
csrc/lstm_pers.hip's launches under hipGraph replay — MODES=gg showed NaN gradients about once in 500 steps while the
flags were cleared by a memset node.)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import dvae_amd
from dvae_amd import ops
from oracle.fill import fill_state_dict, synthetic_eps, synthetic_pair

B, T, N = int(os.environ.get("B", 4)), int(os.environ.get("T", 64)), int(os.environ.get("N", 150))
ops.LSTM_PERSISTENT = os.environ.get("PERS", "1") == "1"
MODES = os.environ.get("MODES", "eg")          # e = eager, g = graph: which kind each of the two trainers is


def make(graph):
    w = dvae_amd.ConvolutionalMulVAE("VCTK", T, 80, 32, 0.0, 0.01, 500, False, batch_size=B, speaker_size=4,
                                     device=torch.device("cuda"), latent_dim=32, mse_cof=10, kl_cof=10)
    w.model.load_state_dict(fill_state_dict(w.model.state_dict()))
    w.model.train()
    w.enable_graph(graph)
    return w


a, b = make(MODES[0] == "g"), make(MODES[1] == "g")
groups = {"enc": ("enc_", "style.", "content."), "dec_lstm2": ("dec_lstm2.",), "dec_other": ("dec_pre", "dec_lstm1", "dec_modules", "dec_linear2"),
          "postnet": ("postnet.",)}
idx = {}
o = a.optimizer
for gname, pre in groups.items():
    sel = [(o.offsets[n], p.numel()) for n, p in zip(o.names, o.params) if n.startswith(pre)
           and not (n.endswith(".0.conv.bias") or (n.startswith("dec_modules.") and n.endswith(".0.bias")))]
    idx[gname] = torch.cat([torch.arange(lo, lo + n, device="cuda") for lo, n in sel])
b1 = o.betas[0]
prev = [a.optimizer.exp_avg.clone(), b.optimizer.exp_avg.clone()]
worst = {g: 0.0 for g in groups}
out = []
for i in range(N):
    x1, x2 = (t.cuda() for t in synthetic_pair(B, T, 100 + i % 5))
    eps = synthetic_eps(B, seed=200 + i % 5)
    a.model.eps_override = eps
    b.model.eps_override = eps
    la, lb = a.step(x1, x2, None, train=True), b.step(x1, x2, None, train=True)
    ga = (a.optimizer.exp_avg - b1 * prev[0]) / (1 - b1)
    gb = (b.optimizer.exp_avg - b1 * prev[1]) / (1 - b1)
    prev = [a.optimizer.exp_avg.clone(), b.optimizer.exp_avg.clone()]
    row = {g: float((ga[ix] - gb[ix]).norm() / gb[ix].norm()) for g, ix in idx.items()}
    dl = max(abs(p - q) / max(abs(q), 1e-9) for p, q in zip(la, lb))
    for g in row:
        worst[g] = max(worst[g], row[g])
    if max(row.values()) > 1e-2 or dl > 1e-5:
        out.append((i, f"loss {dl:.1e}", {g: f"{v:.1e}" for g, v in row.items()}))
print(f"modes={MODES} persistent={ops.LSTM_PERSISTENT}: {N} alternating steps; worst relative gradient difference per group:",
      {g: f"{v:.1e}" for g, v in worst.items()}, "| outlier steps:", out[:8])
