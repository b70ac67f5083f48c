"""Whole-path parity of the HIP DisentangledVAE / ConvolutionalMulVAE (through the C ABI) against
(a) the golden vectors recorded from the real reference and (b) the CPU oracle on the same seeded
inputs.  Tolerance on the 8 loss scalars: 1e-4 relative (BASELINE.json north_star)."""
import os

import numpy as np
import pytest
import torch

from oracle.dvae_ref import RefTrainer, loss_gvae2
from oracle.fill import fill_state_dict, synthetic_eps, synthetic_pair

pytestmark = pytest.mark.gpu
LOSS_RTOL = 1e-4
FW = ["recons_x1", "recons_x2", "recons_x1_hat", "recons_x2_hat", "q_z1_mu", "q_z1_logvar",
      "q_z2_mu", "q_z2_logvar", "z_style_mu", "z_style_logvar"]


def make(batch, n_frames, lr=1e-4):
    import dvae_amd
    w = dvae_amd.ConvolutionalMulVAE("VCTK", n_frames, 80, 32, lr, 0.01, 500, False, batch_size=batch,
                                     speaker_size=4, device=torch.device("cuda"), latent_dim=32, mse_cof=10,
                                     kl_cof=10)
    w.model.load_state_dict(fill_state_dict(w.model.state_dict()))
    assert w.optimizer.views_intact()
    w.model.train()
    return w


def rel(a, b):
    return abs(a - b) / max(1e-12, abs(b))


def is_prebn_conv_bias(name):
    """Conv biases feeding a training-mode BatchNorm: their gradient is mathematically zero, so both the
    reference and this path only produce round-off there (|g| ~ 1e-3 against 1e+3 for real gradients)."""
    return name.endswith(".0.conv.bias") or (name.startswith("dec_modules.") and name.endswith(".0.bias"))


@pytest.mark.parametrize("name", ["c0_b4_t64", "b3_t64", "b2_t128", "c1_b64_t128"])
def test_against_reference_golden(golden_dir, name):
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    B, T = int(g["batch"]), int(g["n_frames"])
    w = make(B, T)
    assert [n for n, _ in w.model.named_parameters()] == list(g["param_names"])
    x1, x2 = (t.cuda() for t in synthetic_pair(B, T, int(g["seed"])))
    w.model.eps_override = tuple(torch.from_numpy(g[k]) for k in ("eps_c1", "eps_c2", "eps_s"))
    w.optimizer.zero_grad()
    outs = w.model(x1, x2)
    losses = w.loss_functionGVAE2(x1, x2, *outs, train=True)
    got = [float(v) for v in torch.stack([l.detach() for l in losses]).tolist()]
    for i, (a, b) in enumerate(zip(got, g["losses_fwd"])):
        assert rel(a, b) <= LOSS_RTOL, f"loss[{i}] {a} vs reference {b}"
    for n, t in zip(FW, outs):
        t = t.detach().cpu()
        if "fw_" + n in g.files:
            np.testing.assert_allclose(t.numpy(), g["fw_" + n], rtol=2e-3, atol=2e-4)
        else:
            assert rel(float(t.double().abs().sum()), float(g["fw_" + n + "_abs"])) <= 1e-4
            np.testing.assert_allclose(t[:, ::16, ::8].numpy(), g["fw_" + n + "_slice"], rtol=5e-3, atol=5e-4)
    losses[0].backward()
    gn = np.array([float(p.grad.double().norm()) for _, p in w.model.named_parameters()])
    ref = g["grad_norm"]
    big = np.array([not is_prebn_conv_bias(n) for n in g["param_names"]])
    # Round 6: no atomics on the step, so these norms are run-to-run bit-identical; what is left against the reference is
    # the ReLU-kink sensitivity of the model itself (one pre-activation within 1e-6 of zero flips its mask and moves the
    # gradients upstream of it by up to 3.6e-3: scripts/dz_diag.py) — the style head sits at the end
    # of the longest such chain.
    style = np.array([n.startswith("style.linear_layer.") for n in g["param_names"]])
    np.testing.assert_allclose(gn[big & ~style], ref[big & ~style], rtol=5e-3)
    np.testing.assert_allclose(gn[style], ref[style], rtol=5e-3)
    assert np.all(gn[~big] < 0.2) and np.all(ref[~big] < 0.2)   # round-off only; real gradient norms are >= 50
    for k in g.files:
        if k.startswith("g_"):
            p = dict(w.model.named_parameters())[k[2:]]
            # relative L2 (ReLU-gate / |.|-sign flips make single elements differ by ~1% of the largest
            # one between any two fp32 implementations, see test_against_oracle_b8_t64_full_gradients)
            err = float(np.linalg.norm(p.grad.cpu().numpy().astype(np.float64) - g[k]))
            # (6e-3: the deepest BatchNorm of the encoder measures 5.03e-3 at B = 64, T = 128 — the same value in every run now —
            # ReLU-kink flips upstream of it, see above; everything else is below 4e-3)
            assert err <= 6e-3 * float(np.linalg.norm(g[k])), (k, err / float(np.linalg.norm(g[k])))
        if k.startswith("bn_"):
            v = w.model.state_dict()[k[3:]].cpu().numpy()
            np.testing.assert_allclose(v, g[k], rtol=2e-4, atol=1e-5, err_msg=k)


@pytest.mark.parametrize("name", ["c0_b4_t64", "b2_t128", "c1_b64_t128"])
def test_two_steps_against_reference_golden(golden_dir, name):
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    B, T = int(g["batch"]), int(g["n_frames"])
    w = make(B, T)
    x1, x2 = (t.cuda() for t in synthetic_pair(B, T, int(g["seed"])))
    w.model.eps_override = tuple(torch.from_numpy(g[k]) for k in ("eps_c1", "eps_c2", "eps_s"))
    s1 = w.step(x1, x2, None, train=True)
    w.model.eps_override = tuple(torch.from_numpy(g[k]) for k in ("eps2_c1", "eps2_c2", "eps2_s"))
    s2 = w.step(x2, x1, None, train=True)
    for i in range(8):
        assert rel(s1[i], g["step1"][i]) <= LOSS_RTOL, (i, s1[i], g["step1"][i])
    # The second step runs on Adam-updated weights.  Adam's first update is lr*sign(g) per weight, so every gradient
    # element that is zero up to round-off moves its weight by +-lr on either side independently: how far that moves the
    # REFERENCE from itself is measured, not argued — tests/golden/trajectory_c0_b4_t64.npz holds its trajectory at four
    # thread counts (step 2: <= 1.1e-7 on the L1 terms, <= 4e-6 on the KL terms; test_trajectory_... below holds all 20
    # steps to 3 x that spread).  Here, on other shapes and one realisation of the reference: 1e-4 on the L1 terms (the
    # single-step contract), 1e-3 on the KL terms.
    print(name, "step 2 relative distances:", [f"{rel(s2[i], g['step2'][i]):.1e}" for i in range(8)])
    # (the report-only style KL — four dimensions at the end of the longest chain — measures 1.4e-3 at B = 64 / T = 128 and
    # 2.4e-3 at B = 2 / T = 128, the same in every run now; the trajectory fixture exists at the c0 shape only, where it is
    # inside 3 x the reference's own perturbed spread)
    for i in range(8):
        assert rel(s2[i], g["step2"][i]) <= (LOSS_RTOL if i < 5 else 1e-3 if i < 7 else 5e-3), (i, s2[i], g["step2"][i])
    pn = np.array([float(p.detach().double().norm()) for _, p in w.model.named_parameters()])
    np.testing.assert_allclose(pn, g["param_norm_after2"], rtol=1e-3, atol=2e-3)


def test_trajectory_inside_the_references_own_spread(golden_dir):
    """20 training steps of the HIP path (default arithmetic) against the REAL reference's trajectory
    (/root/reference/model/variational_base_vae.py:58-70 called 20 times on five cycled input pairs, noise recorded):
    every loss of every step within max(1e-4, 3 x the distance the reference keeps from ITSELF at 1 / 2 / 4 / 8 threads)
    (conftest.trajectory_band) — the multi-step pin that replaces the argued step-2 tolerances."""
    from conftest import trajectory_band
    g = np.load(os.path.join(golden_dir, "trajectory_c0_b4_t64.npz"))
    ref, band = trajectory_band(g)
    B, T = int(g["batch"]), int(g["n_frames"])
    w = make(B, T, lr=float(g["lr"]))
    inputs = [tuple(t.cuda() for t in synthetic_pair(B, T, int(s))) for s in g["input_seeds"]]
    worst = []
    for s in range(int(g["n_steps"])):
        w.model.eps_override = tuple(torch.from_numpy(g[k][s]) for k in ("eps_c1", "eps_c2", "eps_s"))
        x1, x2 = inputs[s % len(inputs)]
        got = np.array(w.step(x1, x2, None, train=True))
        d = np.abs(got - ref[s]) / np.maximum(1e-12, np.abs(ref[s]))
        worst.append(float((d / band[s]).max()))
        assert np.all(d <= band[s]), (s, d, band[s])
    print("trajectory: worst distance / band per step:", np.array2string(np.array(worst), precision=2))


def test_against_oracle_b8_t64_full_gradients():
    """Every parameter gradient, elementwise, against the CPU oracle (B=8, T=64)."""
    B, T = 8, 64
    w = make(B, T)
    tr = RefTrainer(B, n_frames=T)
    tr.model.load_state_dict(fill_state_dict(tr.model.state_dict()))
    tr.model.train()
    x1, x2 = synthetic_pair(B, T, 77)
    eps = synthetic_eps(B, seed=5)
    outs_ref = tr.model(x1, x2, eps)
    l_ref = loss_gvae2(x1, x2, outs_ref, B)
    l_ref[0].backward()
    w.model.eps_override = eps
    w.optimizer.zero_grad()
    outs = w.model(x1.cuda(), x2.cuda())
    l = w.loss_functionGVAE2(x1.cuda(), x2.cuda(), *outs, train=True)
    l[0].backward()
    for i in range(8):
        assert rel(float(l[i].detach()), float(l_ref[i].detach())) <= LOSS_RTOL, (i, float(l[i].detach()), float(l_ref[i].detach()))
    for a, b in zip(outs, outs_ref):
        scale = max(1e-6, float(b.abs().max()))
        assert float((a.detach().cpu() - b.detach()).abs().max()) <= 2e-3 * scale
    # Relative L2 error per parameter.  ReLU gates / |.| signs flip on elements that are zero up to round-off,
    # so two correct fp32 implementations differ by ~1e-3 here (the fp32 oracle is 1.2e-3 away from an fp64
    # run of itself; this path measures 4.5e-4 against the fp32 oracle — scripts/diag_grads.py).
    ref_params = dict(tr.model.named_parameters())
    bad = []
    for n, p in w.model.named_parameters():
        gr = ref_params[n].grad.double()
        err = float((w.model.reference_layout(n, p.grad).cpu().double() - gr).norm())
        if is_prebn_conv_bias(n):
            if err > 0.2:
                bad.append((n, err))
        elif err > 5e-3 * float(gr.norm()):
            bad.append((n, err, float(gr.norm())))
    assert not bad, bad


def test_full_size_properties_b64_t128():
    """BASELINE config[1] size (B=64, T=128): properties that need no CPU oracle run.
    * finite losses, loss decreases over a few Adam steps on a fixed batch;
    * linearity of the loss normalisation: L1 terms scale with 1/batch_size (configured), KL terms do not;
    * pair symmetry: swapping x1<->x2 (and the eps) swaps the per-utterance losses."""
    B, T = 64, 128
    w = make(B, T)
    x1, x2 = (t.cuda() for t in synthetic_pair(B, T, 1234))
    eps = synthetic_eps(B, seed=3)
    w.model.eps_override = eps
    with torch.no_grad():
        w.model.eval()     # eval BatchNorm so the symmetry check is exact w.r.t. running-stat updates
        outs = w.model(x1, x2)
        la = [float(v) for v in w.loss_functionGVAE2(x1, x2, *outs)]
        w.model.eps_override = (eps[1], eps[0], eps[2])
        outs_s = w.model(x2, x1)
        lb = [float(v) for v in w.loss_functionGVAE2(x2, x1, *outs_s)]
    assert rel(la[1], lb[2]) < 1e-5 and rel(la[2], lb[1]) < 1e-5 and rel(la[3], lb[4]) < 1e-5
    assert rel(la[5], lb[6]) < 1e-5
    w.batch_size = 32
    with torch.no_grad():
        lc = [float(v) for v in w.loss_functionGVAE2(x2, x1, *outs_s)]
    assert rel(lc[1], 2 * lb[1]) < 1e-6 and rel(lc[5], lb[5]) < 1e-6
    w.batch_size = B
    w.model.train()
    w.model.eps_override = eps
    hist = [w.step(x1, x2, None, train=True)[0] for _ in range(4)]
    assert all(np.isfinite(hist)) and hist[-1] < hist[0], hist


def test_encode_decode_postnet_api():
    """The standalone entry points voice conversion uses (variational_base_vae.py:278-293), eval mode."""
    import dvae_amd
    B, T = 3, 64
    w = make(B, T)
    tr = RefTrainer(B, n_frames=T)
    tr.model.load_state_dict(fill_state_dict(tr.model.state_dict()))
    w.model.eval()
    tr.model.eval()
    x1, _ = synthetic_pair(B, T, 5)
    with torch.no_grad():
        a = w.model.encode(x1.cuda())
        b = tr.model.encode(x1)
        for u, v in zip(a, b):
            np.testing.assert_allclose(u.cpu().numpy(), v.numpy(), rtol=2e-3, atol=2e-4)
        z = torch.cat((b[0], b[2]), -1)
        r, r_ref = w.model.decode(z.cuda()), tr.model.decode(z)
        np.testing.assert_allclose(r.cpu().numpy(), r_ref.numpy(), rtol=2e-3, atol=2e-4)
        p, p_ref = w.model.postnet(r_ref.cuda()), tr.model.postnet(r_ref)
        np.testing.assert_allclose(p.cpu().numpy(), p_ref.numpy(), rtol=2e-3, atol=2e-4)


def test_cpu_input_fails_loudly():
    w = make(2, 64)
    x1, x2 = synthetic_pair(2, 64, 1)
    with pytest.raises(RuntimeError):
        w.model(x1, x2)


def test_graph_replay_matches_eager():
    """The hipGraph-captured step (device-resident Adam counter, static input/eps buffers) reproduces the eager one.
    lr = 0 keeps the weights fixed, so both trainers see identical states and must agree to round-off on every
    step (with lr > 0 two runs of the SAME code drift apart through Adam's sign(g) updates of ~0 gradients);
    Adam moments, BatchNorm running statistics and counters still evolve and are compared."""
    B, T = 4, 64
    a, b = make(B, T, lr=0.0), make(B, T, lr=0.0)
    b.enable_graph(True)
    for i in range(4):
        x1, x2 = (t.cuda() for t in synthetic_pair(B, T, 100 + i))
        eps = synthetic_eps(B, seed=200 + i)
        a.model.eps_override = eps
        b.model.eps_override = eps
        la = a.step(x1, x2, None, train=True)
        lb = b.step(x1, x2, None, train=True)
        for k in range(8):
            assert rel(lb[k], la[k]) <= 1e-5, (i, k, la[k], lb[k])
    assert b._graph is not None and b.optimizer.t == 4 and a.optimizer.t == 4
    for x, y in ((a.optimizer.exp_avg, b.optimizer.exp_avg), (a.optimizer.exp_avg_sq, b.optimizer.exp_avg_sq)):
        # gradient flip noise: two runs of ONE code path differ by 0.5-2.5e-2 per step in relative gradient norm at this size
        # (scripts/two_trainer_stress.py, any mode), so the moments after 4 steps by up to ~2e-2
        assert float((x - y).norm()) <= 4e-2 * float(x.norm())
    for (n1, v1), (n2, v2) in zip(a.model.named_buffers(), b.model.named_buffers()):
        if n1.endswith("num_batches_tracked"):
            assert int(v1) == int(v2) == 8
        else:
            assert float((v1 - v2).abs().max()) <= 1e-4 * max(1.0, float(v1.abs().max())), n1
    # a manual zero_grad + backward in between (gradients left behind, other launches in the same buffers) does not leak into
    # the replayed step
    x1, x2 = (t.cuda() for t in synthetic_pair(B, T, 150))
    eps = synthetic_eps(B, seed=250)
    b.model.eps_override = eps
    b.optimizer.zero_grad()
    b.loss_functionGVAE2(x1, x2, *b.model(x1, x2), train=True)[0].backward()
    assert float(b.optimizer.flat_g.abs().max()) > 0.0
    a.model.eps_override = eps
    la, lb = a.step(x1, x2, None, train=True), b.step(x1, x2, None, train=True)
    for k in range(8):
        assert rel(lb[k], la[k]) <= 1e-5, (k, la[k], lb[k])
    assert float((a.optimizer.exp_avg - b.optimizer.exp_avg).norm()) <= 4e-2 * float(a.optimizer.exp_avg.norm())
    # and with a real learning rate the replayed steps train: loss goes down
    c = make(B, T)
    c.enable_graph(True)
    x1, x2 = (t.cuda() for t in synthetic_pair(B, T, 7))
    hist = [c.step(x1, x2, None, train=True)[0] for _ in range(5)]
    assert hist[-1] < hist[0], hist


def test_replayed_steps_do_not_depend_on_the_previous_input():
    """lr = 0, four inputs cycled through a graph-replayed trainer: every step on input k must return input k's eight
    losses (to the 1e-6 of the atomically accumulated sums) whatever the step before it saw.  Regression test of the
    persistent LSTM launches' flag clear: as a memset node of the replayed graph it let about one step in a hundred read
    the PREVIOUS step's rows out of the exchange ring (loss off by 1e-5) — invisible when every step sees the same input."""
    B, T, NIN, N = 4, 64, 4, 120
    w = make(B, T, lr=0.0)
    w.enable_graph(True)
    inputs = [tuple(t.cuda() for t in synthetic_pair(B, T, 300 + k)) + (synthetic_eps(B, seed=400 + k),) for k in range(NIN)]
    rows = []
    for i in range(N):
        x1, x2, eps = inputs[i % NIN]
        w.model.eps_override = eps
        rows.append(w.step(x1, x2, None, train=True))
    rows = np.array(rows, dtype=np.float64)
    for k in range(NIN):
        r = rows[k::NIN]
        med = np.median(r, axis=0)
        dev = np.abs(r - med) / np.maximum(np.abs(med), 1e-9)
        assert dev.max() <= 5e-6, (k, float(dev.max()), int(np.argmax(dev.max(axis=1))))


def test_config0_run_training_and_resume(tmp_path):
    """BASELINE configs[0] plumbing on the GPU path: 2 synthetic speakers x 8 utterances [80,96] float64 .npy,
    B=4, T=64 through SpeechDatasetGVAE -> DataLoader -> run_training (hipGraph replay) -> checkpoint -> resume."""
    from dvae_amd import train as cli
    from dvae_amd.data import write_synthetic_corpus
    root = write_synthetic_corpus(str(tmp_path / "corpus"), n_speakers=2, n_utt=8, length=96, seed=0)
    log_dir = str(tmp_path / "results")
    argv = ["--train", "true", f"--dataset_fp={root}", "--batch-size=4", "--latent-size=32", "--speaker_size=4",
            "--lr=1e-4", "--epochs=4", "--report-interval=2", "--mse_cof=10", "--kl_cof=10", f"--log_dir={log_dir}",
            "--samples_length=64", "--seed=3"]
    hist = cli.main(argv)
    assert [h["epoch"] for h in hist] == [1, 2, 3, 4]
    assert all(np.isfinite(list(h.values())).all() for h in hist)
    assert hist[-1]["Loss/Reconstruction Loss1"] < hist[0]["Loss/Reconstruction Loss1"]
    ck = sorted(os.listdir(os.path.join(log_dir, "checkpoints")))
    assert "DisentangledVAE_VCTK_2.pth" in ck and "DisentangledVAE_VCTK_4.pth" in ck
    sd = torch.load(os.path.join(log_dir, "checkpoints", "DisentangledVAE_VCTK_4.pth"))
    assert "enc_lstm.weight_hh_l1_reverse" in sd and "dec_modules.2.0.weight" in sd     # reference key names
    # reconstructions of a test batch at every report interval (variational_base_vae.py:196-201); files carry the
    # value load_last_model returns, i.e. checkpoint epoch + 1, as in the reference (:205-206)
    est = os.listdir(os.path.join(log_dir, "images", "estimation"))
    for ep in (3, 5):
        assert f"{ep}_original_mel_0.npy" in est and f"{ep}_recons_mel_0.npy" in est, est
    # Adam moments are checkpointed per parameter in the REFERENCE's layout (conv weights [Cout][Cin][5])
    osd = torch.load(os.path.join(log_dir, "checkpoints", "DisentangledVAE_VCTK_4.opt"))
    assert osd["format"] == 2 and tuple(osd["exp_avg"]["dec_modules.2.0.weight"].shape) == (512, 512, 5)
    assert "cuda_rng_offset" in osd
    hist2 = cli.main(argv[:7] + ["--epochs=1"] + argv[8:])                               # resumes at epoch 5
    assert [h["epoch"] for h in hist2] == [5]


@pytest.mark.parametrize("B,T", [(2, 256), (1, 512)])
def test_long_segments_against_oracle(B, T):
    """The frame counts of BASELINE configs[2] / configs[4] (T = 256 / 512, fp32 here): losses vs the CPU oracle."""
    w = make(B, T)
    tr = RefTrainer(B, n_frames=T)
    tr.model.load_state_dict(fill_state_dict(tr.model.state_dict()))
    tr.model.train()
    x1, x2 = synthetic_pair(B, T, 31)
    eps = synthetic_eps(B, seed=32)
    with torch.no_grad():
        l_ref = loss_gvae2(x1, x2, tr.model(x1, x2, eps), B)
    w.model.eps_override = eps
    got = w.step(x1.cuda(), x2.cuda(), None, train=True)
    for i in range(8):
        assert rel(got[i], float(l_ref[i])) <= LOSS_RTOL, (i, got[i], float(l_ref[i]))


def test_conversion_path(golden_dir):
    """Inference row: chunking, eval-mode encode/decode/postnet, style swap, clamp — HIP path vs the
    vectors recorded from the real reference and vs the oracle."""
    from oracle.dvae_ref import RefDVAE, convert_mel_ref
    from dvae_amd.model.variational_base_vae import chunking_mel
    g = np.load(os.path.join(golden_dir, "conversion_t64.npz"))
    w = make(4, 64)
    w.model.load_state_dict(fill_state_dict(w.model.state_dict(), salt=3, random_running_stats=True))
    ch = chunking_mel(g["target"], 64)
    assert tuple(ch.shape) == tuple(g["trg_chunks_shape"]) and float(ch[-1].abs().sum()) == 0.0
    out = w.convert_mel(g["source"], g["target"])
    np.testing.assert_allclose(out["source"].cpu().numpy(), g["source_cat"], rtol=0, atol=1e-7)
    for k in ("recons", "converted"):
        err = float(np.abs(out[k].cpu().numpy() - g[k]).max())
        assert err <= 2e-3 * max(1.0, float(np.abs(g[k]).max())), (k, err)
    m = RefDVAE(4, 32, 64)
    m.load_state_dict(fill_state_dict(m.state_dict(), salt=3, random_running_stats=True))
    ref = convert_mel_ref(m, torch.from_numpy(g["source"]), torch.from_numpy(g["target"]))
    d_hip, d_ref = out["spectral_detail"].cpu(), ref["spectral_detail"]
    ok = ref["converted"] > 1e-3                       # the ratio is ill-conditioned where the clamp hits 0
    assert float((d_hip[ok] - d_ref[ok]).abs().max()) <= 5e-2 * float(d_ref[ok].abs().max())
    assert w.model.training                             # convert_mel restores the mode


def test_gpu_pair_loader_matches_dataset_semantics(tmp_path):
    """Input-pipeline row: device-resident corpus + HIP gather/crop vs the CPU dataset rules."""
    from dvae_amd.data import GpuPairLoader, SpeechDatasetGVAE, write_synthetic_corpus
    root = write_synthetic_corpus(str(tmp_path / "corpus"), n_speakers=3, n_utt=6, length=96, seed=1)
    short = np.random.RandomState(2).uniform(0, 1, size=(80, 40))          # shorter than the crop: zero padding
    np.save(os.path.join(root, "spk000", "utt000_mel.npy"), short)
    ds = SpeechDatasetGVAE(root, samples_length=64, seed=4)
    loader = GpuPairLoader(ds, batch_size=3, seed=5)
    assert len(loader) == len(ds) // 3
    files = sorted(loader.index, key=loader.index.get)
    seen = 0
    for x1, x2, spk in loader:
        u1, u2, o1, o2 = loader.last_meta
        assert x1.shape == (3, 80, 64) and x1.is_cuda and spk.shape == (3,)
        for x, us, os_ in ((x1, u1, o1), (x2, u2, o2)):
            for i, (u, o) in enumerate(zip(us, os_)):
                mel = np.load(files[u]).astype(np.float32)
                ref = np.zeros((80, 64), dtype=np.float32)
                part = mel[:, o:o + 64]
                ref[:, :part.shape[1]] = part
                np.testing.assert_array_equal(x[i].cpu().numpy(), ref)
                assert 0 <= o <= max(0, mel.shape[1] - 64)
        for i in range(3):   # same-speaker pairs, label = speaker directory index
            d1, d2 = os.path.dirname(files[u1[i]]), os.path.dirname(files[u2[i]])
            assert d1 == d2 and int(spk[i]) == ds.speaker_ids.index(os.path.basename(d1))
        seen += 1
    assert seen == len(loader)
    w = make(3, 64)
    vals = w.train(loader, 1, logging_func=lambda *a: None)     # the trainer consumes it like a DataLoader
    assert all(np.isfinite(v) for v in vals)


def test_gpu_pair_loader_rank_sharding(tmp_path):
    """Data-parallel sharding of the device-resident loader: ranks take disjoint pairs of the same epoch permutation,
    together they cover every pair once, and every rank sees the same number of batches."""
    from dvae_amd.data import GpuPairLoader, SpeechDatasetGVAE, write_synthetic_corpus
    root = write_synthetic_corpus(str(tmp_path / "corpus"), n_speakers=4, n_utt=8, length=80, seed=1)
    seen = []
    for rank in range(2):
        ds = SpeechDatasetGVAE(root, samples_length=64, seed=4)
        loader = GpuPairLoader(ds, batch_size=2, seed=5, rank=rank, world_size=2)
        assert len(loader) == len(ds) // 2 // 2
        mine = []
        for x1, x2, spk in loader:
            u1, u2, _, _ = loader.last_meta
            mine += list(zip(u1.tolist(), u2.tolist()))
        assert len(mine) == len(loader) * 2
        seen.append(set(mine))
    assert not (seen[0] & seen[1]) and len(seen[0] | seen[1]) == 16
    with pytest.raises(ValueError):
        GpuPairLoader(ds, batch_size=2, rank=0, world_size=2)          # no shared seed


def test_estimate_trained_model(tmp_path):
    """variational_base_vae.py:205-239: last checkpoint -> eval-mode forward(train=False) of one batch -> mels on disk.
    The reconstruction must equal the oracle's eval-mode forward on the same weights."""
    from oracle.dvae_ref import RefDVAE
    w = make(3, 64)
    sd = fill_state_dict(w.model.state_dict(), salt=5, random_running_stats=True)
    ck = tmp_path / "ck"
    ck.mkdir()
    torch.save(sd, str(ck / "DisentangledVAE_VCTK_7.pth"))
    x1, x2 = synthetic_pair(3, 64, 17)
    loader = [(x1, x2, torch.zeros(3, dtype=torch.long))]
    eps = synthetic_eps(3, seed=1)
    w.model.eps_override = eps            # the style noise is drawn even with train=False (disentangled_vae.py:261)
    epoch, r1, r2 = w.estimate_trained_model(loader, str(ck), str(tmp_path / "est"))
    assert epoch == 8 and w.model.training
    files = sorted(os.listdir(tmp_path / "est"))
    assert "8_original_mel_0.npy" in files and "8_recons_mel_2.npy" in files
    m = RefDVAE(4, 32, 64)
    m.load_state_dict(sd)
    m.eval()
    with torch.no_grad():
        ref = m(x1, x2, (None, None, eps[2]))
    got = np.load(tmp_path / "est" / "8_recons_mel_1.npy")
    np.testing.assert_allclose(got, ref[2][1].numpy(), atol=2e-3)
    np.testing.assert_allclose(r2.cpu().numpy(), ref[3].numpy(), atol=2e-3)
