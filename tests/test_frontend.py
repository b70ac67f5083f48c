"""Mel front-end row.  CPU: the numpy restatement against vectors recorded from the imported
reference (tests/golden/frontend.npz) and against properties of the two restated third-party pieces.
GPU: the HIP front-end (through the C ABI) against the restatement."""
import os

import numpy as np
import pytest
import torch

from oracle import mel_ref


@pytest.fixture(scope="module")
def g(golden_dir):
    return np.load(os.path.join(golden_dir, "frontend.npz"))


def test_reference_arithmetic_pinned(g):
    """_amp_to_db / _normalize / _denormalize / lws_num_frames / lws_pad_lr of the real reference."""
    np.testing.assert_allclose(mel_ref.amp_to_db(g["amp"]), g["db"], rtol=0, atol=1e-12)
    np.testing.assert_allclose(mel_ref.normalize(g["sdb"]), g["norm"], rtol=0, atol=1e-15)
    np.testing.assert_allclose(mel_ref.denormalize(g["denorm_in"]), g["denorm"], rtol=0, atol=1e-12)
    for n, m, (l, r) in zip(g["lengths"], g["frames"], g["pads"]):
        assert mel_ref.lws_num_frames(int(n)) == int(m)
        assert mel_ref.lws_pad_lr(int(n)) == (int(l), int(r))
    sr, fft, hop, nm, fmin, fmax, mindb, refdb = g["hparams"]
    assert (sr, fft, hop, nm) == (mel_ref.SAMPLE_RATE, mel_ref.FFT_SIZE, mel_ref.HOP_SIZE, mel_ref.NUM_MELS)
    assert (fmin, fmax, mindb, refdb) == (mel_ref.FMIN, mel_ref.FMAX, mel_ref.MIN_LEVEL_DB, mel_ref.REF_LEVEL_DB)


def test_melspectrogram_glue_pinned(g):
    """melspectrogram() of the real reference with lws / librosa answered by the restatements: same array."""
    mel = mel_ref.melspectrogram(g["wav"].astype(np.float64))
    assert mel.shape == g["mel"].shape == (80, mel_ref.lws_num_frames(len(g["wav"])))
    np.testing.assert_allclose(mel, g["mel"], rtol=0, atol=1e-12)
    assert mel.min() >= 0.0 and mel.max() <= 1.0 and mel.max() > 0.5


def test_restated_lws_window_is_a_perfect_reconstruction_pair():
    """lws uses the same window for analysis and synthesis: sum_m w^2(n - m*hop) must be 1 (what fixes the scale)."""
    w = mel_ref.lws_window()
    ola = sum(np.roll(np.concatenate((w ** 2, np.zeros(3 * 1024))), m * 256) for m in range(8))
    np.testing.assert_allclose(ola[1024:2048], 1.0, atol=1e-12)
    # frames cover the padded signal exactly
    for n in (1, 255, 256, 257, 5000):
        l, r = mel_ref.lws_pad_lr(n)
        assert l + n + r == (mel_ref.lws_num_frames(n) - 1) * 256 + 1024
    # a unit-amplitude sinusoid at a bin centre peaks there with magnitude A * sum(w) / 2
    t = np.arange(8192)
    d = np.abs(mel_ref.lws_stft(np.sin(2 * np.pi * 100 * t / 1024.0)))
    assert d[10].argmax() == 100 and abs(d[10, 100] - w.sum() / 2) < 1e-5 * w.sum()


def test_restated_mel_basis_properties():
    """Slaney construction: triangles between 90 and 7600 Hz, unit area in Hz, peaks increasing in frequency."""
    w = mel_ref.mel_basis()
    assert w.shape == (80, 513) and (w >= 0).all()
    f = np.linspace(0, 8000, 513)
    assert w[:, f < 90].sum() == 0 and w[:, f > 7600].sum() == 0
    peaks = w.argmax(1)
    assert (np.diff(peaks) > 0).all()
    area = w.sum(1) * (f[1] - f[0])                       # ~1 where the triangle spans many bins
    assert np.all(np.abs(area[40:] - 1.0) < 0.05)
    assert abs(float(mel_ref._mel_to_hz(mel_ref._hz_to_mel(3210.0))) - 3210.0) < 1e-9
    assert abs(float(mel_ref._hz_to_mel(1000.0)) - 15.0) < 1e-12


# ------------------------------------------------------------------------------------------------ GPU
def _signal(n, seed):
    rs = np.random.RandomState(seed)
    t = np.arange(n) / 16000.0
    return (0.25 * np.sin(2 * np.pi * (200 + 37 * seed) * t) + 0.1 * np.sin(2 * np.pi * 2500 * t * (1 + 0.3 * t))
            + 0.03 * rs.standard_normal(n)).astype(np.float32)


@pytest.mark.gpu
@pytest.mark.parametrize("n", [1, 255, 256, 1000, 1024, 16000, 16001, 48013])
def test_hip_frontend_matches_restatement(n):
    from dvae_amd.frontend import MelFrontend
    fe = MelFrontend()
    wav = _signal(n, n % 7)
    got = fe.melspectrogram(wav).cpu().numpy()
    ref = mel_ref.melspectrogram(wav.astype(np.float64))
    assert got.shape == ref.shape == (80, fe.num_frames(n))
    # normalised units: 1e-4 = 0.01 dB.  fp32 DFT round-off only matters in bins ~100 dB below the frame's peak.
    assert np.abs(got - ref).max() <= 2e-4, np.abs(got - ref).max()


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["bf16", "fp32"])
def test_hip_frontend_does_not_follow_the_training_compute_mode(mode):
    """Features must be those of the fp32 corpus whatever arithmetic the TRAINING contractions run in: the front-end
    pins its two contractions to exact fp32 products (bf16 operands would put a leakage floor ~50 dB under each frame's
    peak against a 100 dB normalisation range)."""
    from dvae_amd import ops
    from dvae_amd.frontend import MelFrontend
    fe = MelFrontend()
    wav = _signal(16001, 3)
    base = fe.melspectrogram(wav)
    with ops.compute_dtype(mode):
        other = fe.melspectrogram(wav)
        assert ops.get_compute_dtype() == mode
    # not bit-equal run to run (split-K partial sums meet in atomics), but far inside what bf16 operands would do (~1e-2)
    assert float((base - other).abs().max()) <= 2e-6
    ref = mel_ref.melspectrogram(wav.astype(np.float64))
    assert np.abs(other.cpu().numpy() - ref).max() <= 2e-4


@pytest.mark.gpu
def test_hip_frontend_golden_and_batch(g):
    """The vector recorded through the reference's melspectrogram(), and a ragged batch == one call per waveform."""
    from dvae_amd.frontend import MelFrontend
    fe = MelFrontend()
    got = fe.melspectrogram(g["wav"]).cpu().numpy()
    assert np.abs(got - g["mel"]).max() <= 2e-4
    wavs = [_signal(n, i) for i, n in enumerate((3000, 256, 20000, 777))]
    batch = fe.melspectrogram_batch(wavs)
    for w, b in zip(wavs, batch):
        one = fe.melspectrogram(w)
        assert b.shape == one.shape and float((b - one).abs().max()) <= 1e-6
    silent = fe.melspectrogram(np.zeros(4000, dtype=np.float32)).cpu().numpy()
    assert (silent == 0.0).all()                           # max(1e-5, 0) -> -100 dB - 16 -> clipped to 0
    with pytest.raises(ValueError):
        fe.melspectrogram(np.zeros(0, dtype=np.float32))


@pytest.mark.gpu
def test_hip_frontend_feeds_the_corpus_format(tmp_path):
    """waveforms -> [80, L] .npy files -> SpeechDatasetGVAE / GpuPairLoader (the format row f-1 consumes)."""
    from dvae_amd.data import GpuPairLoader, SpeechDatasetGVAE
    from dvae_amd.frontend import MelFrontend
    fe = MelFrontend()
    for s in range(2):
        d = tmp_path / f"p{s}"
        d.mkdir()
        for u, mel in enumerate(fe.melspectrogram_batch([_signal(20000 + 500 * u, 10 * s + u) for u in range(4)])):
            np.save(d / f"u{u}_mel.npy", mel.cpu().numpy())
    ds = SpeechDatasetGVAE(str(tmp_path), samples_length=64, seed=1)
    x1, x2, spk = next(iter(GpuPairLoader(ds, batch_size=2, seed=2)))
    assert x1.shape == (2, 80, 64) and float(x1.min()) >= 0.0 and float(x1.max()) <= 1.0
