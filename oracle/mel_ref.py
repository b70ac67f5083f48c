"""TEST INFRASTRUCTURE ONLY — CPU restatement (numpy, float64) of the mel front-end, SURVEY.md §8(f-4).

Follows /root/reference/preprocessing/utils.py:68-73 (`melspectrogram`), :76-103 (`get_hop_size`,
`_lws_processor`, `lws_num_frames`, `lws_pad_lr`), :108-141 (`_linear_to_mel`, `_build_mel_basis`, `_amp_to_db`,
`_normalize`, `_denormalize`) with the hyper-parameters of preprocessing/hparams.py:58-80
(16 kHz, fft 1024, hop 256, 80 mels, fmin 90, fmax 7600, min_level_db -100, ref_level_db 16, clipping allowed).

PARITY STATUS: **partly unpinned.**  Two steps of the path live in third-party packages that are absent from
/root/reference and from this image, and whose versions the reference does not pin (README.md:19-20):
  * `lws.lws(fft_size, hop, mode="speech").stft(y)` — restated from lws's published behaviour (v1.2): analysis
    window sqrt(periodic Hann * 2*hop/fsize) (the scaling that makes analysis == synthesis window a perfect
    reconstruction pair at 75 % overlap: sum_m w^2(n - m*hop) = 1), the signal zero-padded by fsize-hop samples on
    both sides and on the right up to a whole number of hops, one-sided FFT of fsize points.  The reference's own
    helpers `lws_num_frames` / `lws_pad_lr` (utils.py:82-103) state that padding and frame count and ARE pinned
    (tests/golden/frontend.npz).
  * `librosa.filters.mel(sr, n_fft, fmin, fmax, n_mels)` (positional signature => librosa < 0.10: htk=False,
    Slaney area normalisation) — restated from the published Slaney/Auditory-Toolbox construction.
What IS pinned against the imported reference (tests/golden/make_golden.py `frontend`): `_amp_to_db`, `_normalize`,
`_denormalize`, `lws_num_frames`, `lws_pad_lr`, and the glue of `melspectrogram` (order of abs / mel / dB / ref /
normalise, the transpose) with the two third-party calls replaced by the restatements below.
"""
from __future__ import annotations

import numpy as np

SAMPLE_RATE, FFT_SIZE, HOP_SIZE, NUM_MELS = 16000, 1024, 256, 80
FMIN, FMAX, MIN_LEVEL_DB, REF_LEVEL_DB = 90.0, 7600.0, -100.0, 16.0


# ---- lws (third party, restated)
def lws_window(fsize: int = FFT_SIZE, fshift: int = HOP_SIZE) -> np.ndarray:
    n = np.arange(fsize, dtype=np.float64)
    hann = 0.5 - 0.5 * np.cos(2.0 * np.pi * n / fsize)            # periodic ("symmetric=False")
    return np.sqrt(hann * 2.0 * fshift / fsize)


def lws_num_frames(length: int, fsize: int = FFT_SIZE, fshift: int = HOP_SIZE) -> int:
    """utils.py:82-90"""
    pad = fsize - fshift
    if length % fshift == 0:
        return (length + pad * 2 - fsize) // fshift + 1
    return (length + pad * 2 - fsize) // fshift + 2


def lws_pad_lr(length: int, fsize: int = FFT_SIZE, fshift: int = HOP_SIZE):
    """utils.py:93-101"""
    m = lws_num_frames(length, fsize, fshift)
    pad = fsize - fshift
    t = length + 2 * pad
    r = (m - 1) * fshift + fsize - t
    return pad, pad + r


def lws_stft(y: np.ndarray, fsize: int = FFT_SIZE, fshift: int = HOP_SIZE) -> np.ndarray:
    """-> complex [M, fsize//2+1] (time-major, as lws returns it; the reference transposes, utils.py:69)."""
    y = np.asarray(y, dtype=np.float64)
    left, right = lws_pad_lr(len(y), fsize, fshift)
    x = np.concatenate((np.zeros(left), y, np.zeros(right)))
    m = lws_num_frames(len(y), fsize, fshift)
    idx = np.arange(fsize)[None, :] + fshift * np.arange(m)[:, None]
    return np.fft.rfft(x[idx] * lws_window(fsize, fshift)[None, :], axis=1)


# ---- librosa.filters.mel (third party, restated): Slaney mel scale, triangular filters, area normalisation
def _hz_to_mel(f):
    f = np.asarray(f, dtype=np.float64)
    f_sp, min_log_hz = 200.0 / 3.0, 1000.0
    min_log_mel, logstep = min_log_hz / f_sp, np.log(6.4) / 27.0
    lin = f / f_sp
    return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-30) / min_log_hz) / logstep, lin)


def _mel_to_hz(m):
    m = np.asarray(m, dtype=np.float64)
    f_sp, min_log_hz = 200.0 / 3.0, 1000.0
    min_log_mel, logstep = min_log_hz / f_sp, np.log(6.4) / 27.0
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)


def mel_basis(sr: int = SAMPLE_RATE, n_fft: int = FFT_SIZE, n_mels: int = NUM_MELS, fmin: float = FMIN,
              fmax: float = FMAX) -> np.ndarray:
    """-> [n_mels, n_fft//2+1] float64"""
    fft_f = np.linspace(0.0, sr / 2.0, n_fft // 2 + 1)
    mel_f = _mel_to_hz(np.linspace(_hz_to_mel(fmin), _hz_to_mel(fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = mel_f[:, None] - fft_f[None, :]
    w = np.zeros((n_mels, n_fft // 2 + 1))
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        w[i] = np.maximum(0.0, np.minimum(lower, upper))
    w *= (2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels]))[:, None]
    return w


# ---- the reference's own arithmetic (utils.py:127-141)
def amp_to_db(x):
    min_level = np.exp(MIN_LEVEL_DB / 20.0 * np.log(10.0))
    return 20.0 * np.log10(np.maximum(min_level, x))


def normalize(s):
    return np.clip((s - MIN_LEVEL_DB) / -MIN_LEVEL_DB, 0.0, 1.0)


def denormalize(s):
    return np.clip(s, 0.0, 1.0) * -MIN_LEVEL_DB + MIN_LEVEL_DB


def melspectrogram(y: np.ndarray) -> np.ndarray:
    """utils.py:68-73 -> [80, M] float64 in [0, 1]"""
    d = lws_stft(y).T                                     # [513, M]
    s = amp_to_db(mel_basis() @ np.abs(d)) - REF_LEVEL_DB
    return normalize(s)
