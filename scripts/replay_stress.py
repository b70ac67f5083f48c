"""GPU box: is the train step reproducible?  lr = 0, the same input and noise every time: the eight loss scalars of N
steps (graph replay / eager, persistent LSTM on / off) must agree to the noise of the atomically accumulated split-k
sums (~1e-7).  Prints the worst relative deviation from the median per configuration and the outlier steps."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import dvae_amd
from dvae_amd import ops
from oracle.fill import fill_state_dict, synthetic_eps, synthetic_pair

B, T, N = int(os.environ.get("B", 4)), int(os.environ.get("T", 64)), int(os.environ.get("N", 300))
dtype = os.environ.get("DVAE_COMPUTE_DTYPE", "fp32x3")
ops.set_compute_dtype(dtype)


def make():
    w = dvae_amd.ConvolutionalMulVAE("VCTK", T, 80, 32, 0.0, 0.01, 500, False, batch_size=B, speaker_size=4,
                                     device=torch.device("cuda"), latent_dim=32, mse_cof=10, kl_cof=10)
    w.model.load_state_dict(fill_state_dict(w.model.state_dict()))
    w.model.train()
    return w


x1, x2 = (t.cuda() for t in synthetic_pair(B, T, 150))
eps = synthetic_eps(B, seed=250)
for pers in (True, False):
    for graph in (True, False):
        ops.LSTM_PERSISTENT = pers
        w = make()
        w.enable_graph(graph)
        w.model.eps_override = eps
        rows = np.array([w.step(x1, x2, None, train=True) for _ in range(N)], dtype=np.float64)
        med = np.median(rows, axis=0)
        dev = np.abs(rows - med) / np.maximum(np.abs(med), 1e-9)
        bad = np.where(dev.max(axis=1) > 2e-6)[0]
        print(f"persistent={pers} graph={graph}: worst {dev.max():.2e}; {len(bad)} of {N} steps off by > 2e-6:",
              [(int(i), f"{dev[i].max():.1e}") for i in bad[:12]], flush=True)
