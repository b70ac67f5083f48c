"""GPU box: the gradient wrt the OUTPUT of every conv block of encoder and decoder, HIP path against the fp64 oracle (c0
shape), to find where the backward pass of the whole model leaves the fp32 noise floor although every op alone keeps 1e-6
(scripts/op_precision_audit.py).  Per tensor: relative L2, and the same after removing the per-channel mean of the
difference (a row-independent offset per channel shows as a large first and small second number)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.nn.functional as F
import dvae_amd
from dvae_amd import ops
import dvae_amd.model.disentangled_vae as M
from oracle.dvae_ref import RefTrainer, loss_gvae2
from oracle.fill import fill_state_dict, synthetic_pair

ops.set_deterministic(os.environ.get("DET", "0") == "1")
g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "trajectory_c0_b4_t64.npz"))
B, T = int(g["batch"]), int(g["n_frames"])
x1, x2 = synthetic_pair(B, T, int(g["input_seeds"][0]))
eps = tuple(torch.from_numpy(g[k][0]) for k in ("eps_c1", "eps_c2", "eps_s"))

# ---- oracle, fp64, hooks on the block outputs ([B, C, T] per half)
tr = RefTrainer(B, n_frames=T)
tr.model.load_state_dict(fill_state_dict(tr.model.state_dict()))
tr.model.train()
tr.model.double()
ref = {}


def keep(name):
    def h(gr):
        ref.setdefault(name, []).append(gr.detach().clone())
    return h


m = tr.model


def encode(x):
    nb = x.shape[0]
    for i, blk in enumerate(m.enc_modules):
        if i == 2:
            ref.setdefault("xin_enc2", []).append(x.detach().clone())
        x = F.relu(blk(x))
        x.register_hook(keep(f"enc{i}"))
    seq, _ = m.enc_lstm(x.transpose(1, 2))
    seq.register_hook(keep("enc_lstm_out"))
    pre = m.enc_linear(seq.reshape(nb, -1))
    ref.setdefault("feat_pre", []).append(pre.detach().clone())
    feat = F.relu(pre)
    feat.register_hook(keep("dfeat"))
    st, ct = m.style(feat), m.content(feat)
    s, c = m.speaker_size, m.latent_dim - m.speaker_size
    return st[:, :s], st[:, s:], ct[:, :c], ct[:, c:]


def decode(z):
    h = m.dec_pre_linear2(m.dec_pre_linear1(z))
    h = h.view(z.shape[0], -1, 2 * m.dim_neck)
    h, _ = m.dec_lstm1(h)
    h = h.transpose(1, 2)
    h.register_hook(keep("dec_lstm1"))
    for i, blk in enumerate(m.dec_modules):
        h = F.relu(blk(h))
        h.register_hook(keep(f"dec{i}"))
    h, _ = m.dec_lstm2(h.transpose(1, 2))
    return m.dec_linear2(h).transpose(1, 2)


m.encode, m.decode = encode, decode
d = torch.float64
outs = m(x1.to(d), x2.to(d), tuple(e.to(d) for e in eps))
loss_gvae2(x1.to(d), x2.to(d), outs, B)[0].backward()
# hooks fire in reverse call order: x2's decode ran second -> its gradient arrives first
ref_fr = {}
for k, v in ref.items():
    if v[0].dim() != 3:
        continue
    ref_fr[k] = [torch.cat(pair, 0).permute(2, 0, 1).reshape(T * 2 * B, -1) for pair in ((v[0], v[1]), (v[1], v[0]))]

# ---- HIP path, hooks on the outputs of ConvBnActFn / the decoder's first LSTM
got = {}
real_apply = M.ConvBnActFn.apply
names = iter([f"enc{i}" for i in range(3)] + [f"dec{i}" for i in range(3)] + [f"post{i}" for i in range(5)])


class Wrapped:
    @staticmethod
    def apply(*a):
        out = real_apply(*a)
        n = next(names)
        got["args_" + n] = a
        t = out[0] if isinstance(out, tuple) else out
        if t.requires_grad:
            t.register_hook(lambda gr, n=n: got.__setitem__(n, gr.detach().clone()))
        return out


M.ConvBnActFn = Wrapped
real_lin = M.LinearFn.apply
lin_names = iter(["enc_linear", "style", "content", "dec_pre_linear1", "dec_pre_linear2", "dec_linear2"])


class WrappedLin:
    @staticmethod
    def apply(*a):
        out = real_lin(*a)
        n = next(lin_names)
        got["out_" + n] = out.detach().clone()
        if a[0].requires_grad:
            a[0].register_hook(lambda gr, n=n: got.__setitem__("din_" + n, gr.detach().clone()))
        out.register_hook(lambda gr, n=n: got.__setitem__("dout_" + n, gr.detach().clone()))
        return out


M.LinearFn = WrappedLin
w = dvae_amd.ConvolutionalMulVAE("VCTK", T, 80, 32, 1e-4, 0.01, 500, False, batch_size=B, speaker_size=4,
                                 device=torch.device("cuda"), latent_dim=32, mse_cof=10, kl_cof=10)
w.model.load_state_dict(fill_state_dict(w.model.state_dict()))
w.model.train()
w.model.eps_override = eps
w.optimizer.zero_grad()
o = w.model(x1.cuda(), x2.cuda())
w.loss_functionGVAE2(x1.cuda(), x2.cuda(), *o, train=True)[0].backward()
torch.cuda.synchronize()
print(f"deterministic={ops.deterministic()}")
for k in ["dec2", "dec1", "dec0", "enc2", "enc1", "enc0"]:
    a = got[k].double().cpu()
    b = min(ref_fr[k], key=lambda r: float((a - r).norm()))           # whichever half order the hooks fired in
    diff = a - b
    rel = float(diff.norm() / b.norm())
    dm = diff - diff.mean(0, keepdim=True)
    # error by row position inside a frame (segment index) and by frame
    e = diff.reshape(T, 2 * B, -1)
    by_seg = (e.pow(2).sum((0, 2)).sqrt() / b.reshape(T, 2 * B, -1).pow(2).sum((0, 2)).sqrt()).tolist()
    by_frame = (e.pow(2).sum((1, 2)).sqrt() / b.reshape(T, 2 * B, -1).pow(2).sum((1, 2)).sqrt())
    print(f"dz {k}: relL2 {rel:.2e}  after removing the per-channel mean difference {float(dm.norm() / b.norm()):.2e}  "
          f"nonzero frac got {float((a != 0).double().mean()):.3f} ref {float((b != 0).double().mean()):.3f}")
    print("     by segment:", " ".join(f"{v:.1e}" for v in by_seg))
    ce = e.pow(2).sum((0, 1))
    top = torch.argsort(ce, descending=True)[:6]
    print("     by channel: top-6 share of the squared error", [f"ch{int(c)}: {float(ce[c] / ce.sum()):.2f}" for c in top],
          "| of segment 6 alone:", [f"ch{int(c)}: {float(v):.2f}" for c, v in zip(*[torch.argsort(e[:, 6].pow(2).sum(0), descending=True)[:4]] * 1, (e[:, 6].pow(2).sum(0) / e[:, 6].pow(2).sum()).sort(descending=True).values[:4])])
    print("     by frame (first 6, last 6):", " ".join(f"{float(v):.1e}" for v in by_frame[:6]), "...",
          " ".join(f"{float(v):.1e}" for v in by_frame[-6:]))

# ---- the ReLU of enc_linear: masks that differ between the two implementations, and the pre-activations there
pre = torch.cat(ref["feat_pre"], 0)                       # [2B, 2048] (x1 half first: forward order)
feat = got["out_enc_linear"].double().cpu()
mm = (feat > 0) != (pre > 0)
print(f"enc_linear ReLU: {int(mm.sum())} of {mm.numel()} masks differ; |pre-activation| there: "
      f"{[f'{float(v):.2e}' for v in pre[mm].abs()[:8]]}; rows {sorted(set(mm.nonzero()[:, 0].tolist()))}; "
      f"smallest |pre| overall {float(pre.abs().min()):.2e}")
dfeat_ref = min((torch.cat(p_, 0) for p_ in ((ref["dfeat"][0], ref["dfeat"][1]), (ref["dfeat"][1], ref["dfeat"][0]))),
                key=lambda r: float((got["dout_enc_linear"].double().cpu() - r).norm()))
e = got["dout_enc_linear"].double().cpu() - dfeat_ref
print("d(feat) error by segment:", " ".join(f"{float(v):.1e}" for v in e.norm(dim=1) / dfeat_ref.norm(dim=1)))
for n in ("enc_linear", "dec_pre_linear2", "dec_pre_linear1", "style", "content"):
    k = "din_" + n
    if k in got:
        print(f"{k}: shape {tuple(got[k].shape)} finite {bool(torch.isfinite(got[k]).all())}")

# ---- enc_modules.2 alone, on the REAL tensors of this run: its input, its parameters and the gradient that arrived at its
# output, through an fp64 PyTorch replica of the block (conv k5 -> BatchNorm per utterance half -> ReLU)
a = got["args_enc2"]
xin, cw, cb, bw, bb = (t.detach().double().cpu() for t in a[:5])
N = 2 * B
dz = got["enc2"].double().cpu()                                  # [T*N, C]
to_nct = lambda t: t.reshape(T, N, -1).permute(1, 2, 0).contiguous()            # frames -> [N, C, T]
xr = to_nct(xin).requires_grad_()
wr = cw.permute(1, 2, 0).contiguous()                            # packed [5][Cout][Cin] -> [Cout][Cin][5]
outs = []
for sl in (slice(0, B), slice(B, N)):
    y = F.conv1d(xr[sl], wr, cb, padding=2)
    outs.append(torch.relu(F.batch_norm(y, None, None, bw, bb, True, 0.1, 1e-5)))
torch.cat(outs).backward(to_nct(dz))
dx_ref = xr.grad.permute(2, 0, 1).reshape(T * N, -1)
dx_got = got["enc1"].double().cpu()
e = (dx_got - dx_ref).reshape(T, N, -1)
r = dx_ref.reshape(T, N, -1)
print("enc_modules.2 ALONE on this run's tensors: dx relL2", f"{float(e.norm() / r.norm()):.2e}", "by segment:",
      " ".join(f"{float(v):.1e}" for v in e.pow(2).sum((0, 2)).sqrt() / r.pow(2).sum((0, 2)).sqrt()))
g6, r6 = dx_got.reshape(T, N, -1)[:, 6], r[:, 6]
sc = float((g6 * r6).sum() / (r6 * r6).sum())
print(f"segment 6: best scale got ~ {sc:.6f} * ref, residual after scaling {float((g6 - sc * r6).norm() / r6.norm()):.2e}; "
      f"|dz| of segment 6 relative to the others {float(dz.reshape(T, N, -1)[:, 6].norm() / dz.reshape(T, N, -1)[:, 5].norm()):.3f}; "
      f"|dx ref| seg 6 / seg 5 {float(r6.norm() / r[:, 5].norm()):.3e}")


def replica(xin_nct, dz_nct):
    xr_ = xin_nct.clone().requires_grad_()
    o_ = []
    for sl in (slice(0, B), slice(B, N)):
        y_ = F.conv1d(xr_[sl], wr, cb, padding=2)
        o_.append(torch.relu(F.batch_norm(y_, None, None, bw, bb, True, 0.1, 1e-5)))
    torch.cat(o_).backward(dz_nct)
    return xr_.grad.permute(2, 0, 1).reshape(T * N, -1)


xin_o = torch.cat(ref["xin_enc2"], 0)                      # oracle's input of the block, [N, C, T] (x1 half first)
dz_o = min(ref_fr["enc2"], key=lambda r_: float((dz - r_).norm()))
dx_o = min(ref_fr["enc1"], key=lambda r_: float((dx_got - r_).norm()))
xin_h = to_nct(xin)
seg = lambda t_: " ".join(f"{float(v):.1e}" for v in t_)
dxi = (xin_h - xin_o)
print("input of the block, HIP vs oracle, by segment:", seg(dxi.pow(2).sum((1, 2)).sqrt() / xin_o.pow(2).sum((1, 2)).sqrt()))
for name, xi, dzz in (("xin oracle, dz oracle", xin_o, to_nct(dz_o)), ("xin oracle, dz HIP", xin_o, to_nct(dz)),
                      ("xin HIP, dz oracle", xin_h, to_nct(dz_o))):
    d_ = (replica(xi, dzz) - dx_o).reshape(T, N, -1)
    print(f"fp64 replica with {name}: distance to the ORACLE's dx by segment:",
          seg(d_.pow(2).sum((0, 2)).sqrt() / dx_o.reshape(T, N, -1).pow(2).sum((0, 2)).sqrt()))

torch.save({"xin_h": xin_h.float(), "xin_o": xin_o.float(), "xin_d": (xin_h - xin_o).float(), "dz_o": to_nct(dz_o).float(),
            "wr": wr, "cb": cb, "bw": bw, "bb": bb, "dx_o": dx_o.float(), "B": B, "T": T},
           os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "enc2_block.pt"))
