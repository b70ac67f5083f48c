"""GPU box: gradient norms of the B=64, T=128 golden (tests/golden/c1_b64_t128.npz) against the reference's, for the
backward LSTM recurrence on three-plane fragments (default) and on fp32 fragments (DVAE_BWD=f32)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import dvae_amd
from dvae_amd import _lib, derived
from oracle.fill import fill_state_dict, synthetic_pair

if os.environ.get("DVAE_BWD") == "f32":
    orig = derived.lstm_pack_modes
    derived.lstm_pack_modes = lambda mode, H: (orig(mode, H)[0], _lib.MODE_F32 if mode == _lib.MODE_F32X3 else orig(mode, H)[1])
g = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "c1_b64_t128.npz"))
B, T = int(g["batch"]), int(g["n_frames"])
w = dvae_amd.ConvolutionalMulVAE("VCTK", T, 80, 32, 1e-4, 0.01, 500, False, batch_size=B, speaker_size=4,
                                 device=torch.device("cuda:0"), latent_dim=32, mse_cof=10, kl_cof=10)
w.model.load_state_dict(fill_state_dict(w.model.state_dict()))
w.model.train()
x1, x2 = (t.cuda() for t in synthetic_pair(B, T, int(g["seed"])))
w.model.eps_override = tuple(torch.from_numpy(g[k]) for k in ("eps_c1", "eps_c2", "eps_s"))
w.optimizer.zero_grad()
losses = w.loss_functionGVAE2(x1, x2, *w.model(x1, x2), train=True)
losses[0].backward()
names = [n for n, _ in w.model.named_parameters()]
gn = np.array([float(p.grad.double().norm()) for _, p in w.model.named_parameters()])
ref = g["grad_norm"]
rel = np.abs(gn - ref) / np.maximum(ref, 1e-30)
rel[ref < 0.2] = 0.0          # conv biases in front of a training-mode BatchNorm: mathematically zero, round-off on both sides
order = np.argsort(-rel)
print("bwd =", os.environ.get("DVAE_BWD", "x3"))
for i in order[:8]:
    print(f"{names[i]:45s} got {gn[i]:12.5f} ref {ref[i]:12.5f} rel {rel[i]:.2e}")
