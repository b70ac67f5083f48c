"""Diagnostic (GPU box): per-parameter gradient error of the HIP path vs the CPU oracle (fp32 and fp64)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dvae_amd
from oracle.dvae_ref import RefTrainer, loss_gvae2
from oracle.fill import fill_state_dict, synthetic_eps, synthetic_pair

B, T = 8, 64
w = dvae_amd.ConvolutionalMulVAE("VCTK", T, 80, 32, 1e-4, 0.01, 500, False, batch_size=B, speaker_size=4,
                                 device=torch.device("cuda"), latent_dim=32)
w.model.load_state_dict(fill_state_dict(w.model.state_dict()))
x1, x2 = synthetic_pair(B, T, 77)
eps = synthetic_eps(B, seed=5)
def ref(dtype):
    tr = RefTrainer(B, n_frames=T)
    tr.model.load_state_dict(fill_state_dict(tr.model.state_dict()))
    tr.model.to(dtype)
    outs = tr.model(x1.to(dtype), x2.to(dtype), tuple(e.to(dtype) for e in eps))
    l = loss_gvae2(x1.to(dtype), x2.to(dtype), outs, B)
    l[0].backward()
    return {n: p.grad.double() for n, p in tr.model.named_parameters()}, [float(v) for v in l]
g32, l32 = ref(torch.float32)
g64, l64 = ref(torch.float64)
w.model.eps_override = eps
w.optimizer.zero_grad()
outs = w.model(x1.cuda(), x2.cuda())
l = w.loss_functionGVAE2(x1.cuda(), x2.cuda(), *outs, train=True)
l[0].backward()
print("loss hip ", [float(v) for v in l])
print("loss f32 ", l32)
print("loss f64 ", l64)
print(f"{'param':45s} {'|g|':>10s} {'hip-f64':>10s} {'f32-f64':>10s} {'hip-f32':>10s}  (relative L2)")
for n, p in w.model.named_parameters():
    g = p.grad.detach().cpu().double()
    nb = float(g64[n].norm()) + 1e-30
    print(f"{n:45s} {nb:10.3e} {float((g-g64[n]).norm())/nb:10.2e} {float((g32[n]-g64[n]).norm())/nb:10.2e} {float((g-g32[n]).norm())/nb:10.2e}")
