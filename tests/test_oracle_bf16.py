"""CPU checks of the bf16-mode oracle (oracle/bf16_ref.py): with the rounding switched off it must coincide with the
pinned fp32 oracle (so its hand-written LSTM / conv / linear restatement is the same network, forward and backward);
with the rounding on it must sit a bf16-sized distance away."""
import numpy as np
import torch

from oracle import bf16_ref
from oracle.dvae_ref import RefDVAE, loss_gvae2
from oracle.fill import fill_state_dict, synthetic_eps, synthetic_pair


def _run(cls, B=2, T=64):
    torch.manual_seed(0)
    m = cls(4, 32, T)
    m.load_state_dict(fill_state_dict(m.state_dict()))
    m.train()
    x1, x2 = synthetic_pair(B, T, 11)
    losses = loss_gvae2(x1, x2, m(x1, x2, synthetic_eps(B, seed=12)), B)
    losses[0].backward()
    grads = {k: p.grad.clone() for k, p in m.named_parameters()}
    return [float(l) for l in losses], grads


def test_without_rounding_it_is_the_fp32_oracle(monkeypatch):
    monkeypatch.setattr(bf16_ref, "r16", lambda t: t)
    l_ref, g_ref = _run(RefDVAE)
    l_got, g_got = _run(bf16_ref.RefDVAEBf16)
    for a, b in zip(l_got, l_ref):
        assert abs(a - b) <= 2e-6 * max(1.0, abs(b)), (a, b)
    for k in g_ref:
        if ".0.conv.bias" in k or k.endswith(".0.bias"):       # pre-BatchNorm conv biases: pure round-off
            continue
        d = float((g_got[k] - g_ref[k]).norm()) / max(1e-12, float(g_ref[k].norm()))
        assert d <= 2e-3, (k, d)


def test_with_rounding_it_is_bf16_close():
    l_ref, g_ref = _run(RefDVAE)
    l_got, g_got = _run(bf16_ref.RefDVAEBf16)
    d = [abs(a - b) / max(1.0, abs(b)) for a, b in zip(l_got, l_ref)]
    assert 1e-6 < max(d) < 5e-2, d
    k = "dec_lstm2.weight_hh_l1"
    rel = float((g_got[k] - g_ref[k]).norm()) / float(g_ref[k].norm())
    assert 1e-4 < rel < 0.2, rel
    assert bf16_ref.r16(torch.tensor([1.00390625, 1.005859375])).tolist() == [1.0, 1.0078125]   # ties-to-even, RNE


def _autocast_band(golden_dir=None):
    import os
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    g = np.load(os.path.join(here, "bf16_autocast_c0_b4_t64.npz"))
    c = np.load(os.path.join(here, "c0_b4_t64.npz"))
    return g, c


def test_bf16_oracle_is_inside_the_references_own_autocast_band():
    """The one bf16 execution the reference has is its forward + loss under torch.autocast("cpu", dtype=torch.bfloat16);
    tests/golden/make_golden.py recorded it next to the fp32 run of the same weights, inputs and noise
    (bf16_autocast_c0_b4_t64.npz: vectors from the imported reference).  The bf16 compute mode defined here (operands of every
    contraction rounded, fp32 accumulation, fp32 tensors) must be NO FURTHER from the reference's fp32 losses than the
    reference's own autocast run is, loss by loss — it is in fact 1.5 to 170 times closer (autocast also rounds every output
    tensor, which costs the two KL terms 1 %)."""
    g, c = _autocast_band()
    f, a = g["losses_fp32"], g["losses_autocast_bf16"]
    assert np.allclose(f, c["losses_fwd"], rtol=1e-6)                       # the same run as the pinned fp32 golden
    B, T = int(g["batch"]), int(g["n_frames"])
    x1, x2 = synthetic_pair(B, T, int(g["seed"]))
    eps = tuple(torch.tensor(c[k]) for k in ("eps_c1", "eps_c2", "eps_s"))
    m = bf16_ref.RefDVAEBf16(4, 32, T)
    m.load_state_dict(fill_state_dict(m.state_dict()))
    m.train()
    with torch.no_grad():
        got = np.array([float(l) for l in loss_gvae2(x1, x2, m(x1, x2, eps), B)])
    band = np.abs(a - f)
    assert (band > 0).all() and (band / np.abs(f)).max() < 2e-2            # the band is bf16-sized
    assert (np.abs(got - f) <= band).all(), (np.abs(got - f) / np.abs(f), band / np.abs(f))


def test_bf16_oracle_gradients_are_inside_the_references_autocast_band():
    """The same band for the backward pass: the fixture holds, per parameter, how far the reference's own bf16 execution
    (forward + loss + backward under torch.autocast(bf16)) moves that parameter's gradient away from the reference's fp32
    gradient (relative L2; 6e-3 for the last Postnet BatchNorm, 0.28 for the first).  The bf16 compute mode defined here must not
    move any gradient further than that (measured: 0.30 of it in the median, 0.97 at worst)."""
    g, c = _autocast_band()
    B, T = int(g["batch"]), int(g["n_frames"])
    x1, x2 = synthetic_pair(B, T, int(g["seed"]))
    eps = tuple(torch.tensor(c[k]) for k in ("eps_c1", "eps_c2", "eps_s"))
    grads = {}
    for tag, cls in (("fp32", RefDVAE), ("bf16", bf16_ref.RefDVAEBf16)):
        m = cls(4, 32, T)
        m.load_state_dict(fill_state_dict(m.state_dict()))
        m.train()
        loss_gvae2(x1, x2, m(x1, x2, eps), B)[0].backward()
        grads[tag] = {k: p.grad.detach().clone() for k, p in m.named_parameters()}
    band = dict(zip([str(n) for n in g["grad_names"]], g["grad_dist_autocast"]))
    norms = dict(zip([str(n) for n in g["grad_names"]], g["grad_norm_fp32"]))
    checked = 0
    for k, g32 in grads["fp32"].items():
        if ".0.conv.bias" in k or (k.startswith("dec_modules.") and k.endswith(".0.bias")):      # pre-BatchNorm biases: round-off
            continue
        assert abs(float(g32.norm()) - norms[k]) <= 2e-3 * norms[k], k      # the fp32 oracle's gradient IS the reference's
        d = float((grads["bf16"][k] - g32).norm()) / float(g32.norm())
        assert d <= 1.25 * band[k], (k, d, band[k])
        checked += 1
    assert checked >= 70
