"""Pins the oracle (oracle/dvae_ref.py) against vectors produced by running the real
reference (tests/golden/make_golden.py).  CPU only."""
import os

import numpy as np
import pytest
import torch

from oracle.dvae_ref import RefTrainer, loss_gvae2
from oracle.fill import fill_state_dict, synthetic_pair

# c1_b64_t128 is the benchmarked workload itself (BASELINE.json configs[1]); one oracle step is ~10 s on 8 threads
CASES = ["c0_b4_t64", "b3_t64", "b2_t128", "c1_b64_t128"]
FW = ["recons_x1", "recons_x2", "recons_x1_hat", "recons_x2_hat", "q_z1_mu", "q_z1_logvar",
      "q_z2_mu", "q_z2_logvar", "z_style_mu", "z_style_logvar"]


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz"), allow_pickle=False)


def _trainer(g):
    tr = RefTrainer(int(g["batch"]), n_frames=int(g["n_frames"]))
    tr.model.load_state_dict(fill_state_dict(tr.model.state_dict()))
    tr.model.train()
    return tr


@pytest.mark.parametrize("name", CASES)
def test_state_dict_keys_match_reference(golden_dir, name):
    g = _load(golden_dir, name)
    tr = _trainer(g)
    assert [n for n, _ in tr.model.named_parameters()] == list(g["param_names"])
    assert len(list(g["param_names"])) == 84


@pytest.mark.parametrize("name", CASES)
def test_forward_loss_grads(golden_dir, name):
    g = _load(golden_dir, name)
    tr = _trainer(g)
    B, T = int(g["batch"]), int(g["n_frames"])
    x1, x2 = synthetic_pair(B, T, int(g["seed"]))
    eps = tuple(torch.from_numpy(g[k]) for k in ("eps_c1", "eps_c2", "eps_s"))
    outs = tr.model(x1, x2, eps)
    losses = loss_gvae2(x1, x2, outs, B)
    got = np.array([float(v.detach()) for v in losses])
    np.testing.assert_allclose(got, g["losses_fwd"], rtol=2e-6)
    for n, t in zip(FW, outs):
        t = t.detach()
        if "fw_" + n in g.files:
            np.testing.assert_allclose(t.numpy(), g["fw_" + n], rtol=1e-4, atol=1e-5)
        else:
            np.testing.assert_allclose(float(t.double().abs().sum()), float(g["fw_" + n + "_abs"]), rtol=1e-5)
            np.testing.assert_allclose(t[:, ::16, ::8].numpy(), g["fw_" + n + "_slice"], rtol=1e-3, atol=1e-4)
    losses[0].backward()
    gn = np.array([float(p.grad.double().norm()) for _, p in tr.model.named_parameters()])
    ref = g["grad_norm"]
    # conv biases in front of a training-mode BatchNorm have a mathematically zero gradient
    # (pure round-off in the reference): compare those on an absolute scale only.
    big = ref > 1e-3
    np.testing.assert_allclose(gn[big], ref[big], rtol=2e-3)
    assert np.all(gn[~big] < 1e-2)
    for k in g.files:
        if k.startswith("g_"):
            p = dict(tr.model.named_parameters())[k[2:]]
            scale = max(1e-6, float(np.abs(g[k]).max()))
            assert float(np.abs(p.grad.numpy() - g[k]).max()) <= 2e-3 * scale, k
        if k.startswith("bn_"):
            v = tr.model.state_dict()[k[3:]].numpy()
            np.testing.assert_allclose(v, g[k], rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("name", CASES)
def test_two_train_steps(golden_dir, name):
    g = _load(golden_dir, name)
    tr = _trainer(g)
    B, T = int(g["batch"]), int(g["n_frames"])
    x1, x2 = synthetic_pair(B, T, int(g["seed"]))
    eps1 = tuple(torch.from_numpy(g[k]) for k in ("eps_c1", "eps_c2", "eps_s"))
    eps2 = tuple(torch.from_numpy(g[k]) for k in ("eps2_c1", "eps2_c2", "eps2_s"))
    s1 = tr.step(x1, x2, eps1, train=True)
    s2 = tr.step(x2, x1, eps2, train=True)
    np.testing.assert_allclose(np.array(s1), g["step1"], rtol=2e-6)
    # after one Adam step (|update| = lr per weight, sign-driven) allow a looser match
    np.testing.assert_allclose(np.array(s2), g["step2"], rtol=2e-4)
    pn = np.array([float(p.detach().double().norm()) for _, p in tr.model.named_parameters()])
    np.testing.assert_allclose(pn, g["param_norm_after2"], rtol=1e-3, atol=2e-3)


def test_conversion_path_against_reference(golden_dir):
    """Oracle restatement of the tensor part of voice_conversion_mel / chunking_mel vs the real reference."""
    from oracle.dvae_ref import RefDVAE, chunk_mel, convert_mel_ref
    g = _load(golden_dir, "conversion_t64")
    src, trg = torch.from_numpy(g["source"]), torch.from_numpy(g["target"])
    assert tuple(chunk_mel(src.float()).shape) == tuple(g["src_chunks_shape"])
    assert tuple(chunk_mel(trg.float()).shape) == tuple(g["trg_chunks_shape"])     # L % 64 == 0 -> extra zero chunk
    m = RefDVAE(4, 32, 64)
    m.load_state_dict(fill_state_dict(m.state_dict(), salt=3, random_running_stats=True))
    out = convert_mel_ref(m, src, trg)
    np.testing.assert_allclose(out["source"].numpy(), g["source_cat"], rtol=0, atol=0)
    np.testing.assert_allclose(out["recons"].numpy(), g["recons"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(out["converted"].numpy(), g["converted"], rtol=1e-4, atol=1e-5)


def test_oracle_trajectory_inside_the_references_own_spread(golden_dir):
    """20 training steps of the oracle against the REAL reference's own trajectory (variational_base_vae.py:58-70 called 20
    times on five cycled input pairs, its noise recorded): every loss of every step within 3 x the distance the reference
    keeps from ITSELF when only its thread count changes (conftest.trajectory_band), and the first step — before any
    chaotic amplification — at the 2e-6 of the single-step goldens."""
    from conftest import trajectory_band
    g = _load(golden_dir, "trajectory_c0_b4_t64")
    ref, band = trajectory_band(g, perturbed=False)      # the oracle runs the reference's own kernels: the thread spread alone
    tr = _trainer(g)
    B, T = int(g["batch"]), int(g["n_frames"])
    inputs = [synthetic_pair(B, T, int(s)) for s in g["input_seeds"]]
    worst = 0.0
    for s in range(int(g["n_steps"])):
        eps = tuple(torch.from_numpy(g[k][s]) for k in ("eps_c1", "eps_c2", "eps_s"))
        x1, x2 = inputs[s % len(inputs)]
        got = np.array(tr.step(x1, x2, eps, train=True))
        d = np.abs(got - ref[s]) / np.maximum(1e-12, np.abs(ref[s]))
        assert np.all(d <= band[s]), (s, d, band[s])
        if s == 0:
            np.testing.assert_allclose(got, ref[0], rtol=2e-6)
        worst = max(worst, float((d / band[s]).max()))
    assert worst <= 1.0
    # the reference's autocast(bf16) trajectory is recorded beside it: sanity of the fixture itself
    ac = g["traj_autocast_bf16"]
    assert ac.shape == (2, int(g["n_steps"]), 8) and np.all(np.isfinite(ac))
