"""GPU box: the same 6 training steps (fresh trainer, fixed input and noise, lr = 1e-4, graph replay) over and over in one
process: every repetition must give the same loss trajectory (to the noise of the atomically accumulated sums, which
Adam's sign-like first steps amplify to ~1e-4) — a NaN or an outlier is a race."""
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import dvae_amd
from dvae_amd import ops
from oracle.fill import fill_state_dict, synthetic_eps, synthetic_pair

B, T, REPS = int(os.environ.get("B", 4)), int(os.environ.get("T", 64)), int(os.environ.get("REPS", 100))
ops.set_compute_dtype(os.environ.get("DVAE_COMPUTE_DTYPE", "fp32x3"))
ops.LSTM_PERSISTENT = os.environ.get("PERS", "1") == "1"
graph = os.environ.get("GRAPH", "1") == "1"
x1, x2 = (t.cuda() for t in synthetic_pair(B, T, 7))
eps = synthetic_eps(B, seed=9)
keep = []
rows = []
for rep in range(REPS):
    w = dvae_amd.ConvolutionalMulVAE("VCTK", T, 80, 32, 1e-4, 0.01, 500, False, batch_size=B, speaker_size=4,
                                     device=torch.device("cuda"), latent_dim=32, mse_cof=10, kl_cof=10)
    w.model.load_state_dict(fill_state_dict(w.model.state_dict()))
    w.model.train()
    w.enable_graph(graph)
    w.model.eps_override = eps
    rows.append([w.step(x1, x2, None, train=True)[0] for _ in range(6)])
    if rep < 2:
        keep.append(w)            # older trainers (and their graphs) stay alive, as in a test session
rows = np.array(rows)
med = np.median(rows, axis=0)
dev = np.abs(rows - med) / np.abs(med)
nan = int(np.isnan(rows).any(axis=1).sum())
bad = np.where(~(dev.max(axis=1) < 2e-3))[0]
print(f"persistent={ops.LSTM_PERSISTENT} graph={graph}: {REPS} x 6 steps, NaN runs {nan}, runs off by > 2e-3: {len(bad)}, "
      f"worst finite deviation {np.nanmax(dev):.1e}", [(int(i), [f'{v:.6g}' for v in rows[i]]) for i in bad[:4]])
