"""GPU-box: forward / gradient distances between HIP (fp32, bf16 mode) and the oracles (fp32, bf16), B=2, T=64."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dvae_amd
from dvae_amd import ops
from oracle.bf16_ref import RefDVAEBf16
from oracle.dvae_ref import RefDVAE, loss_gvae2
from oracle.fill import fill_state_dict, synthetic_eps, synthetic_pair
B, T = 2, 64
x1, x2 = synthetic_pair(B, T, 21)
eps = synthetic_eps(B, seed=22)
res = {}
for name, cls in (("o32", RefDVAE), ("o16", RefDVAEBf16)):
    m = cls(4, 32, T); m.load_state_dict(fill_state_dict(m.state_dict())); m.train()
    outs = m(x1, x2, eps)
    l = loss_gvae2(x1, x2, outs, B); l[0].backward()
    res[name] = ([o.detach() for o in outs], {k: p.grad.clone() for k, p in m.named_parameters()})
for name in ("fp32", "bf16"):
    ops.set_compute_dtype(name)
    w = dvae_amd.ConvolutionalMulVAE("VCTK", T, 80, 32, 1e-4, 0.01, 500, False, batch_size=B, speaker_size=4,
                                     device=torch.device("cuda"), latent_dim=32, mse_cof=10, kl_cof=10)
    w.model.load_state_dict(fill_state_dict(w.model.state_dict())); w.model.train()
    w.model.eps_override = eps
    w.optimizer.zero_grad()
    outs = w.model(x1.cuda(), x2.cuda())
    ls = w.loss_functionGVAE2(x1.cuda(), x2.cuda(), *outs, train=True); ls[0].backward()
    res["h" + name[-2:]] = ([o.detach().cpu() for o in outs], {k: p.grad.cpu().clone() for k, p in w.model.named_parameters()})
def d(a, b):
    return float((a - b).norm()) / max(1e-12, float(b.norm()))
pairs = [("h32", "o32"), ("h16", "o16"), ("o16", "o32"), ("h16", "o32"), ("h16", "h32")]
print("forward outputs (rel L2; max abs):")
for i, nm in enumerate(("recon1", "recon2", "recon1_hat", "recon2_hat", "q1_mu", "q1_lv")):
    print(f"  {nm:12s}", "  ".join(f"{a}-{b}: {d(res[a][0][i], res[b][0][i]):.2e}/{float((res[a][0][i]-res[b][0][i]).abs().max()):.1e}" for a, b in pairs))
print("gradients (rel L2):")
for k in ("postnet.convolutions.4.1.weight", "postnet.convolutions.4.0.conv.weight", "dec_linear2.linear_layer.weight",
          "dec_linear2.linear_layer.bias", "dec_lstm2.weight_hh_l1", "dec_modules.0.0.weight", "dec_pre_linear1.weight",
          "enc_linear.linear_layer.weight", "enc_modules.0.0.conv.weight"):
    print(f"  {k:42s}", "  ".join(f"{a}-{b}: {d(res[a][1][k], res[b][1][k]):.2e}" for a, b in pairs))
