"""GPU box: where does the HIP path's step-1 gradient differ from the oracle's by more than the oracle's own fp32 rounding?
Adam's first update is lr * g / (|g| + eps): only the SIGN of a gradient element matters, so the second step's losses move
with the elements whose sign differs.  Per parameter: relative L2 error and the fraction of elements whose sign differs,
for (HIP fp32x3 vs oracle fp64) next to (oracle fp32 vs oracle fp64).  c0 shape (B=4, T=64), the trajectory fixture's step 1."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import dvae_amd
from dvae_amd import ops
from oracle.dvae_ref import RefTrainer, loss_gvae2
from oracle.fill import fill_state_dict, synthetic_pair

g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "trajectory_c0_b4_t64.npz"))
B, T = int(g["batch"]), int(g["n_frames"])
x1, x2 = synthetic_pair(B, T, int(g["input_seeds"][0]))
eps = tuple(torch.from_numpy(g[k][0]) for k in ("eps_c1", "eps_c2", "eps_s"))


def oracle(dt):
    tr = RefTrainer(B, n_frames=T)
    tr.model.load_state_dict(fill_state_dict(tr.model.state_dict()))
    tr.model.train()
    if dt == torch.float64:
        tr.model.double()
    outs = tr.model(x1.to(dt), x2.to(dt), tuple(e.to(dt) for e in eps))
    loss_gvae2(x1.to(dt), x2.to(dt), outs, B)[0].backward()
    return {n: p.grad.detach().double() for n, p in tr.model.named_parameters()}


g64, g32 = oracle(torch.float64), oracle(torch.float32)
det = os.environ.get("DET", "0") == "1"
ops.set_deterministic(det)
w = dvae_amd.ConvolutionalMulVAE("VCTK", T, 80, 32, 1e-4, 0.01, 500, False, batch_size=B, speaker_size=4,
                                 device=torch.device("cuda"), latent_dim=32, mse_cof=10, kl_cof=10)
w.model.load_state_dict(fill_state_dict(w.model.state_dict()))
w.model.train()
w.model.eps_override = eps
w.optimizer.zero_grad()
outs = w.model(x1.cuda(), x2.cuda())
w.loss_functionGVAE2(x1.cuda(), x2.cuda(), *outs, train=True)[0].backward()
torch.cuda.synchronize()
sd_ref = dict(RefTrainer(B, n_frames=T).model.named_parameters())
print(f"deterministic={det}  compute={ops.get_compute_dtype()}")
print(f"{'parameter':44s} {'numel':>9s} | HIP-vs-fp64: relL2  signflip | fp32-vs-fp64: relL2  signflip")
tot = [0, 0, 0]
for n, p in w.model.named_parameters():
    gh = w.model.reference_layout(n, p.grad.detach()).double().cpu() if hasattr(w.model, "reference_layout") else p.grad.detach().double().cpu()
    a, b = g64[n], g32[n]
    if gh.shape != a.shape:
        gh = gh.reshape(a.shape)
    nz = a.abs() > 0
    def stats(x):
        rel = float((x - a).norm() / max(1e-30, a.norm()))
        flip = float(((torch.sign(x) != torch.sign(a)) & nz).sum()) / max(1, int(nz.sum()))
        return rel, flip, int(((torch.sign(x) != torch.sign(a)) & nz).sum())
    rh, fh, nh = stats(gh)
    r3, f3, n3 = stats(b)
    tot[0] += nh; tot[1] += n3; tot[2] += int(nz.sum())
    flag = " <<<" if fh > 3 * max(f3, 1e-5) and float(a.norm()) > 1e-3 else ""
    print(f"{n:44s} {a.numel():9d} | {rh:9.2e} {fh:9.2e} | {r3:9.2e} {f3:9.2e}{flag}")
print(f"sign flips in all: HIP {tot[0]} ({tot[0] / tot[2]:.2e}), oracle fp32 {tot[1]} ({tot[1] / tot[2]:.2e}) of {tot[2]} non-zero elements")
