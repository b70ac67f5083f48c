"""bf16 compute mode (BASELINE configs[2]/[4]; `ops.set_compute_dtype("bf16")`): every contraction rounds its
operands to bf16 (nearest-even) and accumulates in fp32.  A product of two bf16 values is exact in fp32, so against a
PyTorch reference fed the SAME rounded operands only the summation order differs: tolerance 2e-5 of the tensor scale
(tighter than the fp32 tests), i.e. a wrong fragment layout or a missed rounding cannot hide.  Against the unrounded
fp32 reference the distance must be of bf16 size (~1e-3..1e-2): that checks the mode is really on."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture()
def ops():
    import dvae_amd  # noqa: F401
    from dvae_amd import ops as o
    o.set_compute_dtype("bf16")
    yield o
    o.set_compute_dtype("fp32")


def dev(t):
    return t.cuda().contiguous()


def r16(t):
    return t.bfloat16().float()


def close(got, ref, rel=2e-5, name=""):
    got, ref = got.detach().cpu().double(), ref.detach().cpu().double()
    scale = max(1e-6, float(ref.abs().max()))
    err = float((got - ref).abs().max())
    assert err <= rel * scale, f"{name}: max err {err:.3e} vs scale {scale:.3e} (rel {err / scale:.2e})"


def rnd(*shape, seed=0, lo=-1.0, hi=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.rand(*shape, generator=g) * (hi - lo) + lo


def to_frames(x):
    N, C, T = x.shape
    return x.permute(2, 0, 1).reshape(T * N, C).contiguous()


def from_frames(y, N, T):
    return y.reshape(T, N, -1).permute(1, 2, 0)


def test_mode_switch(ops):
    assert ops.get_compute_dtype() == "bf16"
    with ops.compute_dtype("fp32"):
        assert ops.get_compute_dtype() == "fp32"
    assert ops.get_compute_dtype() == "bf16"
    with pytest.raises(ValueError):
        ops.set_compute_dtype("fp8")


@pytest.mark.parametrize("M,N,K", [(128, 128, 32), (200, 72, 80), (16384, 512, 128), (8, 2048, 36), (130, 260, 516)])
def test_gemm_nt(ops, M, N, K):
    x, w, b = rnd(M, K, seed=1), rnd(N, K, seed=2), rnd(N, seed=3)
    y = torch.empty(M, N, device="cuda")
    ops.gemm(dev(x), dev(w), y, dev(b), M, N, K, K, K, N, True, True, 1, ops.EPI_STORE, 1)
    close(y, torch.relu(r16(x).double() @ r16(w).double().t() + b.double()), name="bf16 gemm_nt")
    exact = torch.relu(x.double() @ w.double().t() + b.double())
    d = float((y.cpu().double() - exact).abs().max()) / float(exact.abs().max())
    assert 1e-5 < d < 3e-2, d                      # really computed in bf16


@pytest.mark.parametrize("M,N,K,sk", [(128, 2048, 8192, 16), (8, 56, 2048, 4)])
def test_gemm_nt_splitk(ops, M, N, K, sk):
    x, w, b = rnd(M, K, seed=1), rnd(N, K, seed=2), rnd(N, seed=3)
    y = torch.zeros(M, N, device="cuda")
    ops.gemm(dev(x), dev(w), y, dev(b), M, N, K, K, K, N, True, True, 0, ops.EPI_ATOMIC, sk)
    close(y, r16(x).double() @ r16(w).double().t() + b.double(), name="bf16 gemm_splitk")


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (300, 80, 512), (16384, 128, 256), (64, 64, 20)])
def test_gemm_nn(ops, M, N, K):
    dy, w = rnd(M, K, seed=4), rnd(K, N, seed=5)
    dx = torch.empty(M, N, device="cuda")
    ops.gemm(dev(dy), dev(w), dx, None, M, N, K, K, N, N, True, False)
    close(dx, r16(dy).double() @ r16(w).double(), name="bf16 gemm_nn")


@pytest.mark.parametrize("M,N,K,sk", [(256, 128, 512, 1), (80, 512, 1000, 4), (2048, 128, 16384, 8), (8, 2048, 6, 1),
                                      (64, 64, 96, 1)])
def test_gemm_tn(ops, M, N, K, sk):
    dy, x = rnd(K, M, seed=6), rnd(K, N, seed=7)
    base = rnd(M, N, seed=8)
    dw = dev(base.clone())
    ops.gemm(dev(dy), dev(x), dw, None, M, N, K, M, N, N, False, False, 0, ops.EPI_ATOMIC, sk)
    close(dw, base.double() + r16(dy).double().t() @ r16(x).double(), name="bf16 gemm_tn")


@pytest.mark.parametrize("M,N,K", [(128, 64, 64), (72, 200, 260)])
def test_gemm_tt(ops, M, N, K):
    """A row-contiguous ([K][M]), B k-contiguous ([N][K])."""
    a, b = rnd(K, M, seed=9), rnd(N, K, seed=10)
    c = torch.empty(M, N, device="cuda")
    ops.gemm(dev(a), dev(b), c, None, M, N, K, M, K, N, False, True)
    close(c, r16(a).double().t() @ r16(b).double().t(), name="bf16 gemm_tt")


@pytest.mark.parametrize("N,T,Cin,Cout", [(4, 16, 80, 512), (8, 32, 512, 512), (2, 8, 512, 80), (128, 4, 512, 512)])
def test_conv5(ops, N, T, Cin, Cout):
    from dvae_amd._lib import check, lib, ptr, stream
    L = lib()
    x = r16(rnd(N, Cin, T, seed=1)).double().requires_grad_()
    w = r16(rnd(Cout, Cin, 5, seed=2) * 0.1).double().requires_grad_()
    b = rnd(Cout, seed=3)
    gy = r16(rnd(N, Cout, T, seed=4))
    y_ref = F.conv1d(x, w, b.double(), padding=2)
    y_ref.backward(gy.double())
    R = N * T
    xf, wd = dev(to_frames(x.detach().float())), dev(w.detach().float())
    wp = torch.empty(5, Cout, Cin, device="cuda")
    check(L.dvae_conv_pack_w(ptr(wd), ptr(wp), Cout, Cin, stream()), "pack")
    y = torch.empty(R, Cout, device="cuda")
    check(L.dvae_conv5_fwd(ptr(xf), ptr(wp), ptr(dev(b)), ptr(y), R, N, Cin, Cout, stream()), "fwd")
    close(from_frames(y.cpu(), N, T), y_ref, name="bf16 conv_fwd")
    gyf = dev(to_frames(gy))
    dx = torch.empty(R, Cin, device="cuda")
    check(L.dvae_conv5_dgrad(ptr(gyf), ptr(wp), ptr(dx), R, N, Cin, Cout, stream()), "dgrad")
    close(from_frames(dx.cpu(), N, T), x.grad, name="bf16 conv_dgrad")
    dwp = torch.zeros(5, Cout, Cin, device="cuda")
    check(L.dvae_conv5_wgrad(ptr(gyf), ptr(xf), ptr(dwp), R, N, Cin, Cout, 3, stream()), "wgrad")
    dw = torch.zeros(Cout, Cin, 5, device="cuda")
    check(L.dvae_conv_unpack_add_w(ptr(dwp), ptr(dw), Cout, Cin, stream()), "unpack")
    close(dw, w.grad, name="bf16 conv_wgrad")
