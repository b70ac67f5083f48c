"""GPU box: relative-L2 precision audit of every autograd op of the HIP path against fp64 PyTorch on the CPU, at the
trajectory fixture's shapes (B=4: N=8 segments, T=64) and at the benchmark's (N=128, T=128) — forward output, data
gradient and every parameter gradient.  fp32 arithmetic (a different summation order) keeps ~1e-6..3e-5; 1e-3 means an
operand lost its low planes (bf16 rounding) somewhere.  DET=1: deterministic mode."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
import dvae_amd  # noqa
from dvae_amd import ops

ops.set_deterministic(os.environ.get("DET", "0") == "1")
print(f"compute={ops.get_compute_dtype()} deterministic={ops.deterministic()}")
dev = lambda t: t.detach().float().cuda().contiguous()


def rnd(*shape, seed=0, lo=-1.0, hi=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(*shape, generator=g, dtype=torch.float64) * (hi - lo) + lo).float().double()   # fp32-exact values


def rel(got, ref):
    got, ref = got.detach().cpu().double(), ref.detach().cpu().double()
    return float((got - ref).norm() / max(1e-30, float(ref.norm())))


def show(name, pairs):
    worst = max(v for _, v in pairs)
    print(f"{'<<< ' if worst > 2e-4 else '    '}{name:46s} " + "  ".join(f"{k} {v:.1e}" for k, v in pairs), flush=True)


def lstm(N, T, In, H, bidir):
    ref = torch.nn.LSTM(In, H, 1, batch_first=True, bidirectional=bidir).double()
    for p in ref.parameters():
        p.data = p.data.float().double()
    x = rnd(N, T, In, seed=1)
    xr = x.clone().requires_grad_()
    out_ref, _ = ref(xr)
    gy = rnd(N, T, (2 if bidir else 1) * H, seed=2)
    out_ref.backward(gy)
    P = lambda t: torch.nn.Parameter(dev(t))
    names = ["weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0"]
    ps = [P(getattr(ref, n)) for n in names]
    ps += [P(getattr(ref, n + "_reverse")) for n in names] if bidir else [None] * 4
    xf = dev(x.permute(1, 0, 2).reshape(T * N, In)).requires_grad_()
    h = ops.LstmLayerFn.apply(xf, T, N, *ps)
    h.backward(dev(gy.permute(1, 0, 2).reshape(T * N, -1)))
    pairs = [("fwd", rel(h.reshape(T, N, -1).permute(1, 0, 2), out_ref)),
             ("dx", rel(xf.grad.reshape(T, N, In).permute(1, 0, 2), xr.grad))]
    for i, n in enumerate(names):
        pairs.append((n[:9], rel(ps[i].grad, getattr(ref, n).grad)))
        if bidir:
            pairs.append((n[:9] + "_r", rel(ps[4 + i].grad, getattr(ref, n + "_reverse").grad)))
    show(f"LstmLayerFn N={N} T={T} In={In} H={H} bidir={bidir}", pairs)


def conv_block(N, T, Cin, Cout, act):
    from dvae_amd.ops import ConvBnActFn
    R = N * T
    x = rnd(N, Cin, T, seed=1)
    w = rnd(Cout, Cin, 5, seed=2) * 0.05
    w = w.float().double()
    b, gam, bet = rnd(Cout, seed=3), rnd(Cout, seed=4, lo=0.5, hi=1.5), rnd(Cout, seed=5) * 0.1
    gz = rnd(N, Cout, T, seed=7)
    xr, wr, br, gr, ber = (t.clone().requires_grad_() for t in (x, w, b, gam, bet))
    half = N // 2
    outs = []
    for sl in (slice(0, half), slice(half, N)):      # BatchNorm statistics per utterance of the pair (G = 2)
        y = F.conv1d(xr[sl], wr, br, padding=2)
        u = F.batch_norm(y, None, None, gr, ber, True, 0.1, 1e-5)
        outs.append(torch.relu(u) if act == 1 else torch.tanh(u) if act == 2 else u)
    z_ref = torch.cat(outs)
    z_ref.backward(gz)
    to_fr = lambda t: t.permute(2, 0, 1).reshape(T * t.shape[0], t.shape[1])      # [N, C, T] -> [T*N, C]
    P = lambda t: torch.nn.Parameter(dev(t))
    cw = P(w.permute(2, 0, 1))        # packed [5][Cout][Cin]
    cb, bw, bb = P(b), P(gam), P(bet)
    for p in (cw, cb, bw, bb):
        p.grad = torch.zeros_like(p)
    xin = dev(to_fr(x)).requires_grad_()
    rm, rv = torch.zeros(Cout, device="cuda"), torch.ones(Cout, device="cuda")
    nbt = torch.zeros((), dtype=torch.long, device="cuda")
    z = ConvBnActFn.apply(xin, cw, cb, bw, bb, rm, rv, nbt, None, N, 2, act, True, None, None, None, False)
    z.backward(dev(to_fr(gz)))
    show(f"ConvBnActFn N={N} T={T} {Cin}->{Cout} act={act}",
         [("fwd", rel(z, to_fr(z_ref))), ("dx", rel(xin.grad, to_fr(xr.grad))), ("dW", rel(cw.grad, wr.grad.permute(2, 0, 1))),
          ("dgamma", rel(bw.grad, gr.grad)), ("dbeta", rel(bb.grad, ber.grad))])


def linear(M, K, Nout, act):
    x, w, b = rnd(M, K, seed=1), rnd(Nout, K, seed=2) * 0.05, rnd(Nout, seed=3)
    w = w.float().double()
    gy = rnd(M, Nout, seed=4)
    xr, wr, br = (t.clone().requires_grad_() for t in (x, w, b))
    y = F.linear(xr, wr, br)
    y = torch.relu(y) if act else y
    y.backward(gy)
    P = lambda t: torch.nn.Parameter(dev(t))
    wp, bp = P(w), P(b)
    wp.grad, bp.grad = torch.zeros_like(wp), torch.zeros_like(bp)
    xin = dev(x).requires_grad_()
    out = ops.LinearFn.apply(xin, wp, bp, act)
    out.backward(dev(gy))
    show(f"LinearFn M={M} K={K} Nout={Nout} act={act}", [("fwd", rel(out, y)), ("dx", rel(xin.grad, xr.grad)),
                                                            ("dW", rel(wp.grad, wr.grad)), ("db", rel(bp.grad, br.grad))])


for N, T in ((8, 64), (128, 128)):
    print(f"--- N={N} segments, T={T} frames")
    conv_block(N, T, 80, 512, 1)
    conv_block(N, T, 512, 512, 1)
    conv_block(N, T, 512, 512, 2)
    conv_block(N, T, 512, 80, 0)
    lstm(N, T, 512, 64, True)
    lstm(N, T, 128, 64, True)
    lstm(N, T, 128, 512, False)
    lstm(N, T, 512, 1024, False)
    if T == 64:
        lstm(N, T, 1024, 1024, False)
    linear(N, T * 128, 2048, 1)
    linear(N, 2048, 56, 0)
    linear(N, 2048, 8, 0)
    linear(N // 2, 32, 2048, 0)
    linear(N // 2, 2048, T * 128, 0)
    linear(N * T, 1024, 80, 0)
