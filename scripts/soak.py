"""GPU-box: a few hundred replayed train steps on one synthetic batch: finite, decreasing loss (fp32x3 and bf16 mode)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dvae_amd
from dvae_amd import ops
from dvae_amd.data import SyntheticPairs
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
for mode in ("fp32x3", "bf16"):
    ops.set_compute_dtype(mode)
    torch.manual_seed(0)
    w = dvae_amd.ConvolutionalMulVAE("VCTK", 128, 80, 32, 1e-4, 0.01, 500, False, batch_size=64, speaker_size=4,
                                     device=torch.device("cuda"), latent_dim=32, mse_cof=10, kl_cof=10)
    w.model.train()
    w.enable_graph(True)
    x1, x2, spk = SyntheticPairs(64, 128, device="cuda").batch()
    hist = []
    for i in range(steps):
        l = w.step_async(x1, x2, spk)
        if i % 50 == 0 or i == steps - 1:
            hist.append((i, [round(v, 3) for v in l.tolist()[:1] + l.tolist()[5:7]]))
    print(mode, hist)
    assert all(torch.isfinite(p).all() for p in w.model.parameters())
ops.set_compute_dtype("fp32x3")
