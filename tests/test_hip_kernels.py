"""Per-kernel parity of the HIP path (through the C ABI via dvae_amd.ops) against plain PyTorch fp32
on the CPU.  Tolerances: fp32 contraction with a different summation order -> relative 2e-4 of the
tensor's magnitude (north_star: 1e-4 on the loss scalars, checked in test_hip_model.py)."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=["fp32x3", "fp32"])
def ops(request):
    """Every kernel test runs under both fp32 arithmetics: the default ("fp32x3": exact three-way bf16 split on the bf16
    matrix pipe) and the fp32 MFMA ("fp32") — same tolerances."""
    import dvae_amd  # noqa: F401
    from dvae_amd import ops as o
    o.set_compute_dtype(request.param)
    yield o
    o.set_compute_dtype(o.DEFAULT_COMPUTE_DTYPE)


def dev(t):
    return t.cuda().contiguous()


class persistent_lstm:
    """`with persistent_lstm(ops, flag):` — the W_hh-resident one-launch-per-sequence recurrence on / off (off: the
    one-launch-per-frame kernels, which stay the fallback for shapes without a persistent kernel)."""

    def __init__(self, ops, flag):
        self.ops, self.flag = ops, flag

    def __enter__(self):
        self.prev = self.ops.LSTM_PERSISTENT
        self.ops.LSTM_PERSISTENT = self.flag

    def __exit__(self, *exc):
        self.ops.LSTM_PERSISTENT = self.prev
        self.ops.lstm_pers_check()


def close(got, ref, rel=2e-4, name=""):
    got = got.detach().cpu().double()
    ref = ref.detach().cpu().double()
    scale = max(1e-6, float(ref.abs().max()))
    err = float((got - ref).abs().max())
    assert err <= rel * scale, f"{name}: max err {err:.3e} vs scale {scale:.3e} (rel {err / scale:.2e})"


def rnd(*shape, seed=0, lo=-1.0, hi=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.rand(*shape, generator=g) * (hi - lo) + lo


# ------------------------------------------------------------------ GEMM
# (8192, 4096, 1024): 512 tiles of 256x256 -> the 16-wave 256x256x32 instantiation the benchmark's LSTM input
# projections run (gemm.hip launch_gemm `big`)
@pytest.mark.parametrize("M,N,K", [(128, 128, 16), (200, 72, 80), (16384, 512, 128), (8, 2048, 32), (130, 260, 516),
                                   (8192, 4096, 1024)])
@pytest.mark.parametrize("act", [0, 1, 2])
def test_gemm_nt_bias_act(ops, M, N, K, act):
    x, w, b = rnd(M, K, seed=1), rnd(N, K, seed=2), rnd(N, seed=3)
    y = torch.empty(M, N, device="cuda")
    ops.gemm(dev(x), dev(w), y, dev(b), M, N, K, K, K, N, True, True, act, ops.EPI_STORE, 1)
    ref = x.double() @ w.double().t() + b.double()
    ref = [ref, torch.relu(ref), torch.tanh(ref)][act]
    close(y, ref, name="gemm_nt")


@pytest.mark.parametrize("M,N,K,sk", [(128, 2048, 8192, 16), (8, 56, 2048, 4), (256, 128, 1024, 2)])
def test_gemm_nt_splitk_atomic(ops, M, N, K, sk):
    x, w, b = rnd(M, K, seed=1), rnd(N, K, seed=2), rnd(N, seed=3)
    y = torch.zeros(M, N, device="cuda")
    ops.gemm(dev(x), dev(w), y, dev(b), M, N, K, K, K, N, True, True, 0, ops.EPI_ATOMIC, sk)
    close(y, x.double() @ w.double().t() + b.double(), name="gemm_splitk")


@pytest.mark.parametrize("M,N,K,sk", [(128, 2048, 16384, 32), (128, 16384, 2048, 4), (80, 1024, 16384, 64),
                                      (100, 2048, 16384, 32), (128, 4096, 4096, 4)])
@pytest.mark.parametrize("b_kc", [True, False])
def test_gemm_few_rows_splitk(ops, M, N, K, sk, b_kc):
    """k-split products with <= 128 rows against a large weight (the Linear layers over the flattened frames: enc_linear /
    dec_pre_linear2 forward and data gradient, dec_linear2's weight gradient) at the benchmark's sizes: bias, ragged row
    counts, both weight layouts."""
    x, w, b = rnd(M, K, seed=1), rnd(N, K, seed=2), rnd(N, seed=3)
    y = torch.zeros(M, N, device="cuda")
    wd = dev(w) if b_kc else dev(w.t().contiguous())
    ops.gemm(dev(x), wd, y, dev(b), M, N, K, K, K if b_kc else N, N, True, b_kc, 0, ops.EPI_ATOMIC, sk)
    close(y, x.double() @ w.double().t() + b.double(), name="gemm_few_rows")


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (300, 80, 512), (16384, 128, 256)])
def test_gemm_nn_dgrad(ops, M, N, K):
    dy, w = rnd(M, K, seed=4), rnd(K, N, seed=5)
    dx = torch.empty(M, N, device="cuda")
    ops.gemm(dev(dy), dev(w), dx, None, M, N, K, K, N, N, True, False)
    close(dx, dy.double() @ w.double(), name="gemm_nn")
    # accumulate epilogue
    ops.gemm(dev(dy), dev(w), dx, None, M, N, K, K, N, N, True, False, 0, ops.EPI_ACCUM)
    close(dx, 2 * (dy.double() @ w.double()), name="gemm_nn_accum")


# (4096, 1024, 16256, 8): dW_hh of an H=1024 layer at B=64/T=128 -> 16*4*8 = 512 tiles of 256x256 (the 16-wave
# instantiation with atomic split-K); (2048, 16384, 128, 1): the enc_linear / dec_pre_linear2 outer products
@pytest.mark.parametrize("M,N,K,sk", [(256, 128, 512, 1), (80, 512, 1000, 4), (2048, 128, 16384, 8), (8, 2048, 6, 1),
                                      (4096, 1024, 16256, 8), (2048, 16384, 128, 1)])
def test_gemm_tn_wgrad(ops, M, N, K, sk):
    dy, x = rnd(K, M, seed=6), rnd(K, N, seed=7)
    base = rnd(M, N, seed=8)
    dw = dev(base.clone())
    ops.gemm(dev(dy), dev(x), dw, None, M, N, K, M, N, N, False, False, 0, ops.EPI_ATOMIC, sk)
    close(dw, base.double() + dy.double().t() @ x.double(), name="gemm_tn")


@pytest.mark.parametrize("M,N,K,sk,nb", [(256, 64, 16256, 16, 2), (256, 512, 4096, 4, 2), (256, 128, 1024, 1, 4),
                                         (80, 72, 200, 1, 3)])
def test_gemm_batched_equals_separate_launches(ops, M, N, K, sk, nb):
    """dvae_gemm_f32_batched: nb products of one shape in one launch (the two directions' weight gradients of the H = 64
    BiLSTM).  Unsplit, every product is bit-identical to its own dvae_gemm_f32 launch (same tile arithmetic); split,
    both are compared with fp64."""
    dys = [dev(rnd(K, M, seed=10 + b)) for b in range(nb)]
    xs = [dev(rnd(K, N, seed=20 + b)) for b in range(nb)]
    base = [rnd(M, N, seed=30 + b) for b in range(nb)]
    got = [dev(b_.clone()) for b_ in base]
    epi = ops.EPI_ATOMIC if sk > 1 else ops.EPI_ACCUM
    ops.gemm_batched(dys, xs, got, M, N, K, M, N, N, False, False, epi, sk)
    for b in range(nb):
        ref = base[b].double() + dys[b].cpu().double().t() @ xs[b].cpu().double()
        close(got[b], ref, name=f"batched[{b}]")
        if sk == 1:
            one = dev(base[b].clone())
            ops.gemm(dys[b], xs[b], one, None, M, N, K, M, N, N, False, False, 0, ops.EPI_ACCUM, 1)
            assert torch.equal(one, got[b]), b
    # shared B operand (the dW_ih case: both directions contract with the same x)
    got2 = [torch.zeros(M, N, device="cuda") for _ in range(2)]
    ops.gemm_batched(dys[:2], [xs[0], xs[0]], got2, M, N, K, M, N, N, False, False, epi, sk)
    for b in range(2):
        close(got2[b], dys[b].cpu().double().t() @ xs[0].cpu().double(), name=f"batched shared B[{b}]")
    from dvae_amd._lib import lib
    import ctypes as C
    arr = (C.c_void_p * 5)(*[got[0].data_ptr()] * 5)
    assert lib().dvae_gemm_f32_batched(arr, arr, arr, 5, M, N, K, M, N, N, 0, 0, epi, sk, -1, None) == -1   # batch > 4


def test_gemm_rejects_bad_args(ops):
    from dvae_amd._lib import DvaeHipError
    a = torch.zeros(16, 6, device="cuda")
    with pytest.raises(DvaeHipError):
        ops.gemm(a, a, a, None, 16, 16, 6, 6, 6, 16, True, True)   # K % 4 != 0 for k-contiguous operands


# ------------------------------------------------------------------ conv k5 on frame-major rows
def to_frames(x):  # [N,C,T] -> [T*N, C]
    N, Cc, T = x.shape
    return x.permute(2, 0, 1).reshape(T * N, Cc).contiguous()


def from_frames(y, N, T):  # [T*N, C] -> [N,C,T]
    return y.reshape(T, N, -1).permute(1, 2, 0).contiguous()


# (128, 128, 512, 80) / (128, 128, 80, 512): R = 16 384 rows and 80 output columns of the forward pass / the data gradient:
# the k-split, atomically accumulated form of csrc/gemm.hip narrow_conv_split (the benchmark's postnet shapes)
@pytest.mark.parametrize("N,T,Cin,Cout", [(8, 64, 80, 512), (6, 32, 512, 80), (128, 16, 512, 512), (3, 5, 80, 80),
                                          (128, 128, 512, 80), (128, 128, 80, 512)])
def test_conv5_fwd_dgrad_wgrad(ops, N, T, Cin, Cout):
    from dvae_amd._lib import check, lib, ptr, stream
    L = lib()
    x = rnd(N, Cin, T, seed=1).requires_grad_()
    w = (rnd(Cout, Cin, 5, seed=2) * 0.1).requires_grad_()
    b = rnd(Cout, seed=3)
    y_ref = F.conv1d(x, w, b, padding=2)
    gy = rnd(N, Cout, T, seed=4)
    y_ref.backward(gy)

    R = N * T
    xf, wd = dev(to_frames(x.detach())), dev(w.detach())
    wp = torch.empty(5, Cout, Cin, device="cuda")
    check(L.dvae_conv_pack_w(ptr(wd), ptr(wp), Cout, Cin, stream()), "pack")
    close(wp, w.detach().permute(2, 0, 1), rel=0, name="pack")
    y = torch.empty(R, Cout, device="cuda")
    check(L.dvae_conv5_fwd(ptr(xf), ptr(wp), ptr(dev(b)), ptr(y), R, N, Cin, Cout, -1, stream()), "fwd")
    close(from_frames(y.cpu(), N, T), y_ref, name="conv_fwd")

    gyf = dev(to_frames(gy))
    wpt, dx2 = torch.empty(5, Cin, Cout, device="cuda"), torch.empty(R, Cin, device="cuda")
    check(L.dvae_conv_pack_wt(ptr(wd), ptr(wpt), Cout, Cin, stream()), "pack_t")
    close(wpt, w.detach().permute(2, 1, 0), rel=0, name="pack_t")
    check(L.dvae_conv5_dgrad_t(ptr(gyf), ptr(wpt), ptr(dx2), R, N, Cin, Cout, -1, stream()), "dgrad_t")
    close(from_frames(dx2.cpu(), N, T), x.grad, name="conv_dgrad_t")

    dwp = torch.zeros(5, Cout, Cin, device="cuda")
    check(L.dvae_conv5_wgrad(ptr(gyf), ptr(xf), ptr(dwp), R, N, Cin, Cout, 4, -1, stream()), "wgrad")
    dw = torch.zeros(Cout, Cin, 5, device="cuda")
    check(L.dvae_conv_unpack_add_w(ptr(dwp), ptr(dw), Cout, Cin, stream()), "unpack")
    close(dw, w.grad, name="conv_wgrad")


# ------------------------------------------------------------------ BatchNorm(train) + act, per group
@pytest.mark.parametrize("N,T,Cc,G,act", [(8, 64, 512, 2, 1), (6, 32, 80, 2, 0), (4, 16, 512, 1, 2), (128, 8, 512, 2, 2)])
def test_conv_bn_act_block(ops, N, T, Cc, G, act):
    """Whole ConvBnActFn (conv + BN stats/apply + backward) against torch on CPU, two groups = two calls."""
    Cin = 80
    x = rnd(N, Cin, T, seed=1)
    conv = torch.nn.Conv1d(Cin, Cc, 5, padding=2)
    bn = torch.nn.BatchNorm1d(Cc)
    with torch.no_grad():
        conv.weight.copy_(rnd(Cc, Cin, 5, seed=2) * 0.2)
        conv.bias.copy_(rnd(Cc, seed=3) * 0.5 + 1.0)      # a large mean: stresses the variance computation
        bn.weight.copy_(rnd(Cc, seed=4, lo=0.5, hi=1.5))
        bn.bias.copy_(rnd(Cc, seed=5) * 0.2)
    fact = [lambda t: t, torch.relu, torch.tanh][act]
    xr = x.clone().requires_grad_()
    per = N // G
    outs = [fact(bn(conv(xr[g * per:(g + 1) * per]))) for g in range(G)]   # sequential calls, like the reference
    z_ref = torch.cat(outs, 0)
    gz = rnd(N, Cc, T, seed=6)
    z_ref.backward(gz)

    P = lambda t: torch.nn.Parameter(dev(t.detach().clone()))
    # conv weights are handed to the op in the packed layout [5][Cout][Cin] the parameter lives in
    cw, cb, bw, bb = P(conv.weight.permute(2, 0, 1).contiguous()), P(conv.bias), P(torch.ones(Cc)), P(torch.zeros(Cc))
    with torch.no_grad():
        bw.copy_(bn.weight)
        bb.copy_(bn.bias)
    rm, rv = torch.zeros(Cc, device="cuda"), torch.ones(Cc, device="cuda")
    nbt = torch.zeros((), dtype=torch.long, device="cuda")
    xf = dev(to_frames(x)).requires_grad_()
    z = ops.ConvBnActFn.apply(xf, cw, cb, bw, bb, rm, rv, nbt, None, N, G, act, True)
    close(from_frames(z.detach().cpu(), N, T), z_ref, name="bn_fwd")
    close(rm, bn.running_mean, rel=1e-4, name="running_mean")
    close(rv, bn.running_var, rel=1e-4, name="running_var")
    assert int(nbt.item()) == G
    z.backward(dev(to_frames(gz)))
    close(from_frames(xf.grad.cpu(), N, T), xr.grad, rel=5e-4, name="bn_dx")
    close(cw.grad.permute(1, 2, 0), conv.weight.grad, rel=5e-4, name="conv_w.grad")
    close(bw.grad, bn.weight.grad, rel=5e-4, name="bn_w.grad")
    close(bb.grad, bn.bias.grad, rel=5e-4, name="bn_b.grad")
    assert float(cb.grad.abs().max()) < 1e-2 * max(1.0, float(gz.abs().sum()) ** 0.5)  # ~0 (cancels in BN)


def test_bn_residual(ops):
    N, T, Cc = 4, 16, 80
    x = rnd(N, Cc, T, seed=1)
    P = lambda t: torch.nn.Parameter(dev(t))
    w_t = rnd(Cc, Cc, 5, seed=2) * 0.2                               # torch layout [Cout][Cin][5]
    cw, cb = P(w_t.permute(2, 0, 1).contiguous()), P(torch.zeros(Cc))  # packed [5][Cout][Cin]
    bw, bb = P(torch.ones(Cc)), P(torch.zeros(Cc))
    xf = dev(to_frames(x)).requires_grad_()
    nbt = torch.zeros((), dtype=torch.long, device="cuda")
    z = ops.ConvBnActFn.apply(xf, cw, cb, bw, bb, torch.zeros(Cc, device="cuda"), torch.ones(Cc, device="cuda"),
                              nbt, xf, N, 1, 0, True)
    ref_in = x.clone().requires_grad_()
    ref = ref_in + F.batch_norm(F.conv1d(ref_in, w_t, None, padding=2), None, None, training=True)
    close(from_frames(z.detach().cpu(), N, T), ref, name="residual_fwd")
    gz = rnd(N, Cc, T, seed=3)
    z.backward(dev(to_frames(gz)))
    ref.backward(gz)
    close(from_frames(xf.grad.cpu(), N, T), ref_in.grad, rel=5e-4, name="residual_dx")


@pytest.mark.parametrize("act,G,dy16", [(1, 2, False), (0, 1, False), (1, 2, True)])
def test_bn_bwd_from_y_equals_bn_bwd(ops, act, G, dy16):
    """dvae_bn_bwd_from_y (ReLU mask recomputed from Y, Z never read) against dvae_bn_bwd fed the Z of dvae_bn_apply_fwd:
    the same dY, dgamma, dbeta bit for bit (the mask is the sign of the forward pass's own expression); tanh is refused."""
    from dvae_amd._lib import check, lib, ptr, stream
    L = lib()
    R, N, C = 64 * 24, 24, 512
    y = dev(rnd(R, C, seed=1) * 2.0)
    dz = dev(rnd(R, C, seed=2))
    ga, be = dev(rnd(C, seed=3, lo=0.5, hi=1.5)), dev(rnd(C, seed=4) * 0.3)
    mean, rstd = torch.empty(G, C, device="cuda"), torch.empty(G, C, device="cuda")
    ws = torch.empty(L.dvae_bn_ws_bytes(R, C, G), dtype=torch.uint8, device="cuda")
    check(L.dvae_bn_stats_fwd(ptr(y), ptr(mean), ptr(rstd), None, None, None, ptr(ws), R, N, C, G, 1e-5, 0.1, stream()), "stats")
    z = torch.empty(R, C, device="cuda")
    check(L.dvae_bn_apply_fwd(ptr(y), ptr(mean), ptr(rstd), ptr(ga), ptr(be), None, ptr(z), R, N, C, G, act, 0, stream()), "apply")
    if act == 1:
        assert 0.2 < float((z > 0).float().mean()) < 0.8      # the mask matters
    dt = torch.bfloat16 if dy16 else torch.float32
    out = []
    for from_y in (False, True):
        dy = torch.empty(R, C, device="cuda", dtype=dt)
        dg, db = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
        if from_y:
            check(L.dvae_bn_bwd_from_y(ptr(dz), ptr(y), ptr(mean), ptr(rstd), ptr(ga), ptr(be), ptr(dy), ptr(dg), ptr(db),
                                       ptr(ws), R, N, C, G, act, 2 if dy16 else 0, stream()), "from_y")
        else:
            check(L.dvae_bn_bwd(ptr(dz), ptr(y), ptr(z), ptr(mean), ptr(rstd), ptr(ga), ptr(dy), ptr(dg), ptr(db),
                                ptr(ws), R, N, C, G, act, 2 if dy16 else 0, stream()), "bwd")
        out.append((dy.float().cpu(), dg.cpu(), db.cpu()))
    for a, b in zip(*out):
        assert torch.equal(a, b)
    dy = torch.empty(R, C, device="cuda")
    assert L.dvae_bn_bwd_from_y(ptr(dz), ptr(y), ptr(mean), ptr(rstd), ptr(ga), ptr(be), ptr(dy), None, None, ptr(ws),
                                R, N, C, G, 2, 0, stream()) != 0


# ------------------------------------------------------------------ LSTM
# (128, 4, ., 1024) / (128, 4, ., 512): N >= 97 selects the 32-row eight-wave tiles the benchmark runs (lstm.hip
# plan_seq: n_j * ceil(N/32) >= 256); H = 128 / 256 reach the generic LDS-staged frame kernels
@pytest.mark.parametrize("N,T,In,H,bidir", [(8, 12, 512, 64, True), (128, 6, 128, 512, False), (6, 5, 512, 1024, False),
                                            (20, 9, 128, 64, True), (128, 4, 512, 1024, False), (128, 4, 128, 512, True),
                                            (20, 5, 64, 128, False), (20, 5, 96, 256, True), (33, 3, 512, 1024, False),
                                            # H = 64: 4 / 8 / 16 segments per workgroup (N <= 128 / <= 256 / more, two directions)
                                            (160, 3, 128, 64, True), (300, 3, 128, 64, True)])
@pytest.mark.parametrize("persistent", [True, False])
def test_lstm_layer(ops, N, T, In, H, bidir, persistent):
    with persistent_lstm(ops, persistent):
        _lstm_layer(ops, N, T, In, H, bidir)


def _lstm_layer(ops, N, T, In, H, bidir):
    ref = torch.nn.LSTM(In, H, 1, batch_first=True, bidirectional=bidir)
    x = rnd(N, T, In, seed=1)
    xr = x.clone().requires_grad_()
    out_ref, _ = ref(xr)
    gy = rnd(N, T, (2 if bidir else 1) * H, seed=2)
    out_ref.backward(gy)

    P = lambda t: torch.nn.Parameter(dev(t.detach().clone()))
    names = ["weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0"]
    ps = [P(getattr(ref, n)) for n in names]
    ps += [P(getattr(ref, n + "_reverse")) for n in names] if bidir else [None] * 4
    xf = dev(x.permute(1, 0, 2).reshape(T * N, In)).requires_grad_()
    h = ops.LstmLayerFn.apply(xf, T, N, *ps)
    close(h.detach().cpu().reshape(T, N, -1).permute(1, 0, 2), out_ref, name="lstm_fwd")
    h.backward(dev(gy.permute(1, 0, 2).reshape(T * N, -1)))
    close(xf.grad.cpu().reshape(T, N, In).permute(1, 0, 2), xr.grad, rel=5e-4, name="lstm_dx")
    for i, n in enumerate(names):
        close(ps[i].grad, getattr(ref, n).grad, rel=5e-4, name=n)
        if bidir:
            close(ps[4 + i].grad, getattr(ref, n + "_reverse").grad, rel=5e-4, name=n + "_reverse")


@pytest.mark.parametrize("N,T,In,H", [(128, 8, 512, 1024), (6, 4, 512, 1024), (128, 4, 128, 512), (40, 6, 64, 512)])
@pytest.mark.parametrize("persistent", [True, False])
def test_lstm_stack2(ops, N, T, In, H, persistent):
    with persistent_lstm(ops, persistent):
        _lstm_stack2(ops, N, T, In, H)


def _lstm_stack2(ops, N, T, In, H):
    """ops.LstmStack2Fn (two stacked layers sharing frame launches, the dec_lstm2 schedule of the benchmark) against a
    2-layer nn.LSTM.  N = 128 at H = 1024 is the benchmarked instantiation (32-row tiles, 64-deep stacked forward)."""
    assert ops.LstmStack2Fn.usable(T, H, 2, False)
    ref = torch.nn.LSTM(In, H, 2, batch_first=True)
    x = rnd(N, T, In, seed=1)
    xr = x.clone().requires_grad_()
    out_ref, _ = ref(xr)
    gy = rnd(N, T, H, seed=2)
    out_ref.backward(gy)
    P = lambda t: torch.nn.Parameter(dev(t.detach().clone()))
    names = [f"{n}_l{l}" for l in (0, 1) for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")]
    ps = [P(getattr(ref, n)) for n in names]
    xf = dev(x.permute(1, 0, 2).reshape(T * N, In)).requires_grad_()
    h = ops.LstmStack2Fn.apply(xf, T, N, *ps)
    close(h.detach().cpu().reshape(T, N, -1).permute(1, 0, 2), out_ref, name="stack2_fwd")
    h.backward(dev(gy.permute(1, 0, 2).reshape(T * N, -1)))
    close(xf.grad.cpu().reshape(T, N, In).permute(1, 0, 2), xr.grad, rel=5e-4, name="stack2_dx")
    for i, n in enumerate(names):
        close(ps[i].grad, getattr(ref, n).grad, rel=5e-4, name=n)


# ------------------------------------------------------------------ Linear
@pytest.mark.parametrize("M,K,Nout,act", [(8, 8192, 2048, 1), (128, 2048, 56, 0), (8, 32, 2048, 0), (512, 1024, 80, 0)])
def test_linear_fn(ops, M, K, Nout, act):
    x = rnd(M, K, seed=1)
    w, b = rnd(Nout, K, seed=2) * 0.05, rnd(Nout, seed=3)
    xr, wr, br = x.clone().requires_grad_(), w.clone().requires_grad_(), b.clone().requires_grad_()
    y_ref = F.linear(xr, wr, br)
    y_ref = torch.relu(y_ref) if act else y_ref
    gy = rnd(M, Nout, seed=4)
    y_ref.backward(gy)
    xd = dev(x).requires_grad_()
    wd, bd = torch.nn.Parameter(dev(w)), torch.nn.Parameter(dev(b))
    y = ops.LinearFn.apply(xd, wd, bd, act)
    close(y, y_ref, name="linear_fwd")
    y.backward(dev(gy))
    close(xd.grad, xr.grad, rel=5e-4, name="linear_dx")
    close(wd.grad, wr.grad, rel=5e-4, name="linear_dw")
    close(bd.grad, br.grad, rel=5e-4, name="linear_db")


# ------------------------------------------------------------------ latent / KL / L1 / Adam / layout
def test_latent_kl_l1(ops):
    Bh, S, Cn = 5, 4, 28
    style, content = rnd(2 * Bh, 2 * S, seed=1), rnd(2 * Bh, 2 * Cn, seed=2)
    e1, e2, es = rnd(Bh, Cn, seed=3), rnd(Bh, Cn, seed=4), rnd(Bh, S, seed=5)
    sr, cr = style.clone().requires_grad_(), content.clone().requires_grad_()
    s_mu1, s_lv1, s_mu2, s_lv2 = sr[:Bh, :S], sr[:Bh, S:], sr[Bh:, :S].detach(), sr[Bh:, S:].detach()
    smu, slv = (s_mu1 + s_mu2) / 2, (s_lv1 + s_lv2) / 2
    zs = es * torch.exp(0.5 * slv) + smu
    zc1 = e1 * torch.exp(0.5 * cr[:Bh, Cn:]) + cr[:Bh, :Cn]
    zc2 = e2 * torch.exp(0.5 * cr[Bh:, Cn:]) + cr[Bh:, :Cn]
    z_ref = torch.cat((torch.cat((zs, zc1), -1), torch.cat((zs, zc2), -1)), 0)
    q1mu, q1lv = torch.cat((smu, cr[:Bh, :Cn]), -1), torch.cat((slv, cr[:Bh, Cn:]), -1)
    q2mu, q2lv = torch.cat((smu, cr[Bh:, :Cn]), -1), torch.cat((slv, cr[Bh:, Cn:]), -1)
    kl = lambda mu, lv: 1 + lv - mu.pow(2) - lv.exp()
    wz = rnd(2 * Bh, S + Cn, seed=6)
    loss_ref = (z_ref * wz).sum() + (-0.5) * kl(q1mu, q1lv).sum(-1).mean() + (-0.5) * kl(q2mu, q2lv).sum(-1).mean() \
        + 3.0 * (-kl(smu, slv).sum() / 7.0)
    loss_ref.backward()

    sd, cd = dev(style).requires_grad_(), dev(content).requires_grad_()
    z, q_mu, q_lv, s_mu, s_lv = ops.LatentFn.apply(sd, cd, dev(torch.cat((e1, e2), 0)), dev(es), Bh, S, Cn)
    close(z, z_ref, name="z")
    close(q_mu[:Bh], q1mu, name="q1mu")
    close(q_lv[Bh:], q2lv, name="q2lv")
    close(s_mu, smu, name="smu")
    k1 = ops.KlFn.apply(q_mu[:Bh], q_lv[:Bh], -0.5 / Bh)
    k2 = ops.KlFn.apply(q_mu[Bh:], q_lv[Bh:], -0.5 / Bh)
    k3 = ops.KlFn.apply(s_mu, s_lv, -1.0 / 7.0)
    loss = (z * dev(wz)).sum() + k1 + k2 + 3.0 * k3
    close(loss, loss_ref, rel=1e-5, name="latent loss")
    loss.backward()
    close(sd.grad, sr.grad, rel=1e-4, name="dstyle")
    close(cd.grad, cr.grad, rel=1e-4, name="dcontent")
    assert float(sd.grad[Bh:].abs().max()) == 0.0   # x2 style head is detached


@pytest.mark.parametrize("n", [655360, 1003, 4])
def test_l1_sum(ops, n):
    x, y = rnd(n, seed=1, lo=0, hi=1), rnd(n, seed=2)
    yr = y.clone().requires_grad_()
    ref = (x - yr).abs().sum() / 64.0
    (ref * 10.0).backward()
    yd = dev(y).requires_grad_()
    out = ops.L1SumFn.apply(dev(x), yd, 1.0 / 64.0)
    close(out, ref.double(), rel=1e-5, name="l1")
    (out * 10.0).backward()
    close(yd.grad, yr.grad, rel=1e-6, name="l1 grad")


def test_adam_flat_matches_torch(ops):
    from dvae_amd.optim import FlatAdam
    ps = [torch.nn.Parameter(rnd(37, 5, seed=1)), torch.nn.Parameter(rnd(1001, seed=2)), torch.nn.Parameter(rnd(4, 4, seed=3))]
    ref = [torch.nn.Parameter(p.detach().clone()) for p in ps]
    dps = [torch.nn.Parameter(dev(p.detach().clone())) for p in ps]
    opt_ref = torch.optim.Adam(ref, lr=1e-3)
    opt = FlatAdam(dps, lr=1e-3)
    for it in range(3):
        opt.zero_grad()
        for k, (p, r) in enumerate(zip(dps, ref)):
            g = rnd(*p.shape, seed=10 * it + k) * (10.0 ** (k - 1))
            r.grad = g.clone()
            p.grad.add_(dev(g))
        opt_ref.step()
        opt.step()
    for p, r in zip(dps, ref):
        close(p, r, rel=1e-5, name="adam")


def test_layouts(ops):
    from dvae_amd._lib import check, lib, ptr, stream
    Bh, Cc, T = 3, 80, 50
    x1, x2 = rnd(Bh, Cc, T, seed=1), rnd(Bh, Cc, T, seed=2)
    X = ops.mel_to_frames(dev(x1), dev(x2))
    ref = torch.cat((x1, x2), 0).permute(2, 0, 1).reshape(T * 2 * Bh, Cc)
    close(X, ref, rel=0, name="mel_to_frames")
    back = ops.FramesToMelFn.apply(X, 2 * Bh, Cc, T)
    close(back, torch.cat((x1, x2), 0), rel=0, name="frames_to_mel")
    a = rnd(7, 5, 128, seed=3)
    out = ops.Permute102Fn.apply(dev(a), 7, 5, 128, (5, 7, 128))
    close(out, a.permute(1, 0, 2), rel=0, name="permute")
    t = rnd(100, 260, seed=4)
    close(ops.transpose2d(dev(t)), t.t(), rel=0, name="transpose")
    acc = torch.zeros(260, device="cuda")
    ops.colsum_add(dev(t), acc)
    close(acc, t.sum(0), rel=1e-5, name="colsum")
    # bf16 input (the storage of conv / LSTM gradients in the bf16 compute mode): 8 columns per 16-byte load, ragged
    # column counts (80 = the mel width, 520) and row counts that are not a multiple of the block's rows
    for R_, C_ in ((300, 80), (1000, 512), (131, 520)):
        tb = rnd(R_, C_, seed=5).bfloat16()
        acc1, acc2 = torch.zeros(C_, device="cuda"), torch.ones(C_, device="cuda")
        ops.colsum_add(tb.cuda(), acc1, acc2)
        close(acc1, tb.float().sum(0), rel=1e-5, name=f"colsum bf16 {R_}x{C_}")
        close(acc2, tb.float().sum(0) + 1.0, rel=1e-5, name=f"colsum bf16 second output {R_}x{C_}")


def test_prof_hooks(ops):
    ops.prof_enable(1)
    x, w = dev(rnd(256, 64, seed=1)), dev(rnd(128, 64, seed=2))
    y = torch.empty(256, 128, device="cuda")
    for _ in range(3):
        ops.gemm(x, w, y, None, 256, 128, 64, 64, 64, 128, True, True)
    ms, n, fl = ops.prof_collect()
    ops.prof_enable(0)
    assert n == 3 and ms > 0 and fl == 3 * 2.0 * 256 * 128 * 64


# ------------------------------------------------------------------ edge cases / error behaviour
@pytest.mark.parametrize("N,T,In,H,bidir", [(1, 1, 128, 64, True), (17, 3, 128, 512, False), (33, 2, 512, 1024, False)])
@pytest.mark.parametrize("persistent", [True, False])
def test_lstm_ragged_sizes(ops, N, T, In, H, bidir, persistent):
    """Segment counts that are not multiples of the 16-row MFMA tile, single frame, single segment."""
    with persistent_lstm(ops, persistent):
        _lstm_layer(ops, N, T, In, H, bidir)


@pytest.mark.parametrize("N,T", [(1, 1), (1, 3), (2, 2), (5, 7)])
def test_conv_block_tiny(ops, N, T):
    """Fewer rows than one tile; every tap partly or wholly in the zero padding."""
    test_conv5_fwd_dgrad_wgrad(ops, N, T, 80, 80)


def test_gemm_degenerate_shapes(ops):
    for M, N, K in [(1, 4, 4), (3, 8, 36), (128, 4, 16), (7, 132, 20)]:
        x, w = rnd(M, K, seed=1), rnd(N, K, seed=2)
        y = torch.empty(M, N, device="cuda")
        ops.gemm(dev(x), dev(w), y, None, M, N, K, K, K, N, True, True)
        close(y, x.double() @ w.double().t(), name=f"gemm {M}x{N}x{K}")


def test_abi_rejects_bad_arguments_without_crashing():
    """Error behaviour of the C ABI: negative code, no exception across the boundary, no device fault."""
    from dvae_amd._lib import lib, stream
    L = lib()
    a = torch.zeros(64, 64, device="cuda")
    p, st = a.data_ptr(), stream()
    assert L.dvae_gemm_f32(None, p, p, None, 64, 64, 64, 64, 64, 64, 1, 1, 0, 0, 1, -1, st) == -1          # null A
    assert L.dvae_gemm_f32(p + 4, p, p, None, 64, 64, 60, 64, 64, 64, 1, 1, 0, 0, 1, -1, st) == -1         # misaligned
    assert L.dvae_gemm_f32(p, p, p, None, 64, 64, 64, 62, 64, 64, 1, 1, 0, 0, 1, -1, st) == -1             # lda % 4
    assert L.dvae_gemm_f32(p, p, p, None, 64, 64, 64, 64, 64, 64, 1, 1, 1, 2, 4, -1, st) == -1             # act + split-K
    assert L.dvae_gemm_f32(p, p, p, None, 0, 64, 64, 64, 64, 64, 1, 1, 0, 0, 1, -1, st) == -1              # empty
    assert L.dvae_gemm_f32(p, p, p, None, 64, 64, 64, 64, 64, 64, 1, 1, 0, 0, 1, 3, st) == -1               # unknown mode
    assert L.dvae_bn_stats_fwd(p, p, p, None, None, None, p, 64, 8, 63, 2, 1e-5, 0.1, st) == -1        # C % 4
    assert L.dvae_bn_stats_fwd(p, p, p, None, None, None, p, 64, 7, 64, 2, 1e-5, 0.1, st) == -1        # R % N, N % G
    assert L.dvae_lstm_seq_fwd(None, 1, 4, 8, 64, 64, st) == -1
    assert L.dvae_adam_flat(p, p, p, p, 64, 1e-3, 0.9, 0.999, 1e-8, 1.0, 0, st) == -1                  # step < 1
    assert L.dvae_l1_sum_fwd(p, p + 4, p, p, 64, 1.0, st) == -1                                        # misaligned
    assert L.dvae_mel_to_frames(None, None, p, 4, 80, 64, 0, st) == -1
    # the slab forms: a split without slabs, a slab stride shorter than one result, misaligned / ragged sums, bad descriptors
    from dvae_amd._lib import SlabDesc
    assert L.dvae_gemm_f32_slabs(p, p, p, None, 0, 0, None, 64, 64, 64, 64, 64, 64, 1, 1, 0, 4, -1, st) == -1
    assert L.dvae_gemm_f32_slabs(p, p, p, p, 64 * 64 - 4, 4, None, 64, 64, 64, 64, 64, 64, 1, 1, 0, 4, -1, st) == -1
    assert L.dvae_slab_sum(None, p, 64, 2, 64, 0, 0, st) == -1
    assert L.dvae_slab_sum(p, p + 4, 64, 2, 64, 0, 0, st) == -1                                        # misaligned slab
    assert L.dvae_slab_sum(p, p, 62, 2, 64, 0, 0, st) == -1                                            # stride % 4
    assert L.dvae_slab_sum(p, p, 64, 2, 62, 0, 0, st) == -1                                            # n % 4
    assert L.dvae_slab_sum(p, None, 64, 2, 64, 0, 0, st) == -1                                         # slabs named, none given
    assert L.dvae_slab_fold(None, 1, st) == -1
    bad = (SlabDesc * 1)(SlabDesc(p, p, 64, 62, 2, 0))                                                 # n % 4
    assert L.dvae_slab_fold(bad, 1, st) == -1
    bad = (SlabDesc * 1)(SlabDesc(p, None, 64, 64, 2, 0))
    assert L.dvae_slab_fold(bad, 1, st) == -1
    assert L.dvae_slab_fold(bad, 0, st) == 0                                                           # nothing to fold
    assert L.dvae_colsum_add_ws(p, p, None, 64, 64, 64, 0, None, st) == -1                             # no workspace
    assert L.dvae_colsum_add_ws(p, p, None, 64, 64, 62, 0, p, st) == -1                                # ld % 4
    torch.cuda.synchronize()
    assert float(a.abs().sum()) == 0.0


def test_bn_one_group_equals_two_separate_calls(ops):
    """G=2 over a concatenated batch == two G=1 calls on the halves (the reference's two encode() calls)."""
    N, T, Cc = 8, 16, 512
    x = rnd(N, 80, T, seed=1)
    P = lambda t: torch.nn.Parameter(dev(t))
    cw, cb = P((rnd(Cc, 80, 5, seed=2) * 0.2).permute(2, 0, 1).contiguous()), P(rnd(Cc, seed=3))
    bw, bb = P(rnd(Cc, seed=4, lo=0.5, hi=1.5)), P(rnd(Cc, seed=5) * 0.1)
    def run(xs, G, rm, rv, nbt):
        return ops.ConvBnActFn.apply(dev(to_frames(xs)), cw, cb, bw, bb, rm, rv, nbt, None, xs.shape[0], G, 2, True)
    rm2, rv2 = torch.zeros(Cc, device="cuda"), torch.ones(Cc, device="cuda")
    nb2 = torch.zeros((), dtype=torch.long, device="cuda")
    z2 = from_frames(run(x, 2, rm2, rv2, nb2).detach().cpu(), N, T)
    rm1, rv1 = torch.zeros(Cc, device="cuda"), torch.ones(Cc, device="cuda")
    nb1 = torch.zeros((), dtype=torch.long, device="cuda")
    za = from_frames(run(x[:4], 1, rm1, rv1, nb1).detach().cpu(), 4, T)
    zb = from_frames(run(x[4:], 1, rm1, rv1, nb1).detach().cpu(), 4, T)
    close(z2, torch.cat((za, zb), 0), rel=1e-5, name="groups")
    close(rm2, rm1, rel=1e-5, name="running_mean order")
    close(rv2, rv1, rel=1e-5, name="running_var order")
    assert int(nb2) == int(nb1) == 2


@pytest.mark.parametrize("N,T,Cin,Cout,G", [(5, 7, 80, 80, 1), (6, 21, 80, 512, 2), (128, 9, 512, 512, 2), (2, 100, 512, 80, 2),
                                            (66, 3, 80, 512, 2)])
def test_conv_epilogue_bn_statistics_equal_the_standalone_pass(ops, N, T, Cin, Cout, G):
    """dvae_conv5_fwd_stats (BatchNorm partial sums written by the conv epilogue, one fp64 pair per 64-row chunk, group
    and column) + dvae_bn_stats_finalize against dvae_conv5_fwd + dvae_bn_stats_fwd (the separate statistics pass), on
    ragged row counts (partial chunks, chunks straddling a frame, fewer segments than rows per chunk)."""
    from dvae_amd._lib import check, lib, ptr, stream
    L = lib()
    R = N * T
    x = dev(rnd(R, Cin, seed=1))
    wp = dev(rnd(5, Cout, Cin, seed=2) * 0.1)
    b = dev(rnd(Cout, seed=3) + 1.0)
    mode = ops.current_mode()
    out = {}
    for fused in (False, True):
        y = torch.empty(R, Cout, device="cuda")
        mean, rstd = torch.empty(G, Cout, device="cuda"), torch.empty(G, Cout, device="cuda")
        rm, rv = torch.zeros(Cout, device="cuda"), torch.ones(Cout, device="cuda")
        nbt = torch.zeros((), dtype=torch.long, device="cuda")
        ws = torch.full((L.dvae_bn_ws_bytes(R, Cout, G),), 0xFF, device="cuda", dtype=torch.uint8)   # poison: every slot must be written
        if fused:
            check(L.dvae_conv5_fwd_stats(ptr(x), ptr(wp), ptr(b), ptr(y), R, N, Cin, Cout, mode, G, ptr(ws), stream()), "fwd_stats")
            check(L.dvae_bn_stats_finalize(ptr(mean), ptr(rstd), ptr(rm), ptr(rv), ptr(nbt), ptr(ws), R, N, Cout, G, 1e-5, 0.1,
                                           stream()), "finalize")
        else:
            check(L.dvae_conv5_fwd(ptr(x), ptr(wp), ptr(b), ptr(y), R, N, Cin, Cout, mode, stream()), "fwd")
            check(L.dvae_bn_stats_fwd(ptr(y), ptr(mean), ptr(rstd), ptr(rm), ptr(rv), ptr(nbt), ptr(ws), R, N, Cout, G, 1e-5, 0.1,
                                      stream()), "stats")
        out[fused] = (y, mean, rstd, rm, rv, int(nbt))
    assert torch.equal(out[True][0], out[False][0])                    # same conv output
    for k, name in ((1, "mean"), (2, "rstd"), (3, "running_mean"), (4, "running_var")):
        close(out[True][k], out[False][k], rel=2e-5, name=name)
    assert out[True][5] == out[False][5] == G
    # and against torch on the CPU
    yc = out[False][0].cpu().double().reshape(T, N, Cout)
    per = N // G
    for g in range(G):
        seg = yc[:, g * per:(g + 1) * per].reshape(-1, Cout)
        close(out[True][1][g], seg.mean(0), rel=2e-5, name="mean vs torch")
        close(out[True][2][g], 1.0 / torch.sqrt(seg.var(0, unbiased=False) + 1e-5), rel=1e-4, name="rstd vs torch")


@pytest.mark.parametrize("B,T", [(3, 17), (4, 64)])
def test_fused_loss_gvae2(ops, B, T):
    """ops.LossGVAE2Fn (the eight scalars of loss_functionGVAE2 in two launches, one backward launch) against the
    reference's formulas in torch on the CPU (disentangled_vae.py:310-327), with an upstream gradient on EVERY output."""
    n = 80 * T
    x = [rnd(B, 80, T, seed=s, lo=0.0) for s in (1, 2)]
    r = [rnd(B, 80, T, seed=s).requires_grad_() for s in (3, 4, 5, 6)]
    q = [rnd(B, 32, seed=s).requires_grad_() for s in (7, 8, 9, 10)]
    st = [rnd(B, 4, seed=s).requires_grad_() for s in (11, 12)]
    bs, mse_cof, kl_cof = 7.0, 10.0, 3.0
    l1 = [F.l1_loss(r[k], x[k & 1], reduction="sum") / bs for k in range(4)]
    kl = lambda mu, lv: (-0.5) * torch.mean(torch.sum(1 + lv - mu.pow(2) - lv.exp(), dim=-1))
    k1, k2 = kl(q[0], q[1]), kl(q[2], q[3])
    ks = (-1.0) * torch.sum(1 + st[1] - st[0].pow(2) - st[1].exp()) / bs
    loss = mse_cof * (l1[0] + l1[1] + l1[2] + l1[3]) + kl_cof * (k1 + k2)
    ref = torch.stack([loss, *l1, k1, k2, ks])
    gv = rnd(8, seed=13) + 1.5
    (ref * gv).sum().backward()

    d = lambda t: dev(t.detach()).requires_grad_(t.requires_grad)
    xd, rd, qd, sd = [d(t) for t in x], [d(t) for t in r], [d(t) for t in q], [d(t) for t in st]
    out = ops.LossGVAE2Fn.apply(xd[0], xd[1], *rd, *qd, *sd, 1.0 / bs, -0.5 / B, -1.0 / bs, mse_cof, kl_cof)
    close(out, ref, rel=2e-6, name="loss vector")
    (out * dev(gv)).sum().backward()
    for a, b_, nm in zip(rd + qd + sd, r + q + st, ["r1", "r2", "h1", "h2", "q1mu", "q1lv", "q2mu", "q2lv", "smu", "slv"]):
        close(a.grad, b_.grad, rel=1e-5, name="d" + nm)
    # only the total (what the training step differentiates)
    rd2 = [d(t) for t in r]
    out2 = ops.LossGVAE2Fn.apply(xd[0], xd[1], *rd2, *[t.detach() for t in qd], *[t.detach() for t in sd], 1.0 / bs,
                                 -0.5 / B, -1.0 / bs, mse_cof, kl_cof)
    out2[0].backward()
    assert rd2[2].grad is not None and float(rd2[2].grad.abs().max()) == pytest.approx(mse_cof / bs, rel=1e-6)


# ------------------------------------------------------------------ zero / sum (what replaced ATen's fills and gradient adds)
@pytest.mark.parametrize("n", [4, 7, 1000, 4099, 1 << 20])
def test_zero_f32(ops, n):
    from dvae_amd._lib import check, lib, ptr, stream
    x = torch.full((n + 8,), 3.0, device="cuda")
    check(lib().dvae_zero_f32(ptr(x), n, stream()), "dvae_zero_f32")
    assert float(x[:n].abs().max()) == 0.0 and bool((x[n:] == 3.0).all())     # exactly n elements, nothing behind them
    assert lib().dvae_zero_f32(x.data_ptr() + 4, 4, stream()) != 0                # misaligned: refused, not a fault


@pytest.mark.parametrize("n,three", [(4, False), (4096, True), (40960, False), (1 << 20, True)])
def test_sum_f32_and_fanout(ops, n, three):
    from dvae_amd._lib import check, lib, ptr, stream
    a, b, c = (dev(rnd(n, seed=s)) for s in (1, 2, 3))
    out = torch.empty_like(a)
    check(lib().dvae_sum_f32(ptr(a), ptr(b), ptr(c) if three else None, ptr(out), n, stream()), "dvae_sum_f32")
    want = a + b + (c if three else 0)
    assert torch.equal(out, want)                        # same additions in the same order: bit-exact
    # through autograd: a tensor with several consumers gets ONE summed gradient
    x = dev(rnd(n // 4, 4, seed=4)).requires_grad_()
    y = x * 2.0
    parts = ops.fanout(y, 3 if three else 2)
    loss = sum(((k + 1.0) * p).sum() for k, p in enumerate(parts))
    loss.backward()
    k = 6.0 if three else 3.0
    assert torch.equal(x.grad, torch.full_like(x, 2.0 * k))


# ------------------------------------------------------------------ k-split slabs (round 6): no atomics, fixed summation order
def _rel_l2(got, ref):
    got, ref = got.detach().cpu().double(), ref.detach().cpu().double()
    return float((got - ref).norm() / max(1e-30, float(ref.norm())))


@pytest.mark.parametrize("M,N,K,kc,sk", [(128, 2048, 8192, True, 16), (8, 56, 2048, True, 4), (2048, 128, 16384, False, 8),
                                         (4096, 1024, 16384, False, 4), (512, 512, 4096, False, 6), (256, 8, 2048, True, 8)])
def test_gemm_slabs_match_fp64_and_are_bitwise_reproducible(ops, M, N, K, kc, sk):
    """dvae_gemm_f32_slabs + dvae_slab_sum: a product cut along k whose splits are STORED (split 0 into the result, the others
    into slabs) and added in a fixed order — against fp64 in relative L2 (1e-5: an fp32 contraction in another summation order
    keeps ~1e-6; 1e-3 would be a lost low plane of the split arithmetic), with bias and ReLU applied by the sum, twice bit
    for bit (the atomic epilogue this replaces differs from run to run), and equal to the unsplit product to round-off."""
    a, b, bias = rnd(M, K, seed=1), rnd(K, N, seed=2), rnd(N, seed=3)
    A = dev(a if kc else a.t().contiguous())
    B = dev(b.t().contiguous() if kc else b)
    lda, ldb = (K if kc else M), (K if kc else N)
    ref = torch.relu(a.double() @ b.double() + bias.double())
    outs = []
    for _ in range(2):
        Cm = torch.empty(M, N, device="cuda")
        ops.gemm_split(A, B, Cm, dev(bias), M, N, K, lda, ldb, N, kc, kc, 1, sk, None)
        outs.append(Cm)
    assert torch.equal(outs[0], outs[1])
    assert _rel_l2(outs[0], ref) < 1e-5, _rel_l2(outs[0], ref)
    one = torch.empty(M, N, device="cuda")
    ops.gemm(A, B, one, dev(bias), M, N, K, lda, ldb, N, kc, kc, 1, ops.EPI_STORE, 1)
    assert _rel_l2(outs[0], one) < 5e-6      # two fp32 summation orders over K up to 16 384


def test_wgrad_slabs_accumulate_and_fold_like_a_plain_sum(ops):
    """ops.wgrad_gemm on a gradient no optimiser owns: `grad += A^T B` with K cut into slabs, summed right behind the launch;
    called twice it accumulates twice (the semantics of the atomic epilogue it replaces), bit-identically run to run."""
    M, N, K = 1024, 512, 32768
    a, b = rnd(K, M, seed=4), rnd(K, N, seed=5)
    ref = a.double().t() @ b.double()
    res = []
    for _ in range(2):
        g = torch.zeros(M, N, device="cuda")
        ops.wgrad_gemm(dev(a), dev(b), g, None, M, N, K, M, N, False, False, 8, None)
        once = g.clone()
        ops.wgrad_gemm(dev(a), dev(b), g, None, M, N, K, M, N, False, False, 8, None)
        res.append((once, g))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    assert _rel_l2(res[0][0], ref) < 1e-5 and _rel_l2(res[0][1], 2 * ref) < 1e-5


@pytest.mark.parametrize("R,C_,ld,b16", [(16384, 512, 512, False), (65536, 4096, 4096, False), (1000, 80, 80, False),
                                          (8192, 256, 512, False), (512, 2048, 2048, False)])
def test_colsum_is_deterministic_and_exact(ops, R, C_, ld, b16):
    """ops.colsum_add (dvae_colsum_add_ws): per-row-block partial sums, the last workgroup of a column block adds them in
    row-block order and is the one writer — against fp64 and twice bit for bit; it ADDS to what the outputs hold."""
    x = rnd(R, ld, seed=6)
    ref = x[:, :C_].double().sum(0)
    outs = []
    for _ in range(2):
        o1, o2 = torch.ones(C_, device="cuda"), torch.zeros(C_, device="cuda")
        ops.colsum_add(dev(x), o1, o2, rows=R, cols=C_, ld=ld)
        outs.append((o1, o2))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert float((outs[0][1].cpu().double() - ref).abs().max()) <= 2e-6 * float(ref.abs().max() + R ** 0.5)
    assert torch.equal(outs[0][0], outs[0][1] + 1.0) or float((outs[0][0] - outs[0][1] - 1.0).abs().max()) < 1e-3


def test_conv_wgrad_slabs_match_fp64(ops):
    """dvae_conv5_wgrad_slabs (tap mode 2: five per-tap outputs inside every slab) through ops.ConvBnActFn's backward on a
    gradient no optimiser owns: dW against fp64 in relative L2, twice bit for bit."""
    from dvae_amd.ops import ConvBnActFn
    N, T, Cin, Cout = 16, 256, 512, 512
    x, w = rnd(N, Cin, T, seed=1), rnd(Cout, Cin, 5, seed=2) * 0.05
    gz = rnd(N, Cout, T, seed=7)
    xr, wr = x.double().requires_grad_(), w.double().requires_grad_()
    gam, bet = rnd(Cout, seed=4, lo=0.5, hi=1.5), rnd(Cout, seed=5) * 0.1
    y = F.conv1d(xr, wr, None, padding=2)
    torch.relu(F.batch_norm(y, None, None, gam.double(), bet.double(), True, 0.1, 1e-5)).backward(gz.double())
    grads = []
    for _ in range(2):
        P = lambda t: torch.nn.Parameter(dev(t))
        cw, cb, bw, bb = P(w.permute(2, 0, 1)), P(torch.zeros(Cout)), P(gam), P(bet)
        for p_ in (cw, cb, bw, bb):
            p_.grad = torch.zeros_like(p_)
        xin = dev(to_frames(x)).requires_grad_()
        rm, rv = torch.zeros(Cout, device="cuda"), torch.ones(Cout, device="cuda")
        z = ConvBnActFn.apply(xin, cw, cb, bw, bb, rm, rv, torch.zeros((), dtype=torch.long, device="cuda"), None, N, 1, 1,
                              True, None, None, None, False)
        z.backward(dev(to_frames(gz)))
        grads.append((cw.grad.clone(), xin.grad.clone()))
    assert torch.equal(grads[0][0], grads[1][0]) and torch.equal(grads[0][1], grads[1][1])
    assert _rel_l2(grads[0][0], wr.grad.permute(2, 0, 1)) < 1e-5
    assert _rel_l2(grads[0][1], to_frames(xr.grad)) < 1e-5
