"""Mel front-end on the GPU: waveform -> [80, M] normalised log-mel, the array format the training
corpus stores as `<speaker>/*.npy` and `SpeechDatasetGVAE` / `GpuPairLoader` consume.

Mirrors /root/reference/preprocessing/utils.py:68-73 `melspectrogram(y)` with preprocessing/hparams.py:58-80
(16 kHz, fft 1024, hop 256, 80 mels, 90-7600 Hz, min_level_db -100, ref_level_db 16, clipping allowed):

    D = lws.lws(1024, 256, mode="speech").stft(y).T ; S = 20*log10(max(1e-5, mel_basis @ |D|)) - 16
    return clip((S + 100) / 100, 0, 1)

MI355X formulation: every frame of every utterance of a batch is one row of a `[sum M, 1024]` matrix; the STFT is ONE
fp32 contraction against a precomputed `[2*516, 1024]` cos/-sin basis (one-sided spectrum, 513 bins padded to 516 so
that rows stay 16-byte aligned), the mel projection a second one against `[80, 516]`, both on `dvae_gemm_f32` (MFMA);
framing+window, magnitude and dB/normalise/transpose are three HBM-bound HIP passes (`csrc/frontend.hip`).
The window and the two bases are built on the host in float64 (they are constants of the hyper-parameters):
`lws` and `librosa` are NOT dependencies — their published constructions are restated here (see oracle/mel_ref.py
for the parity status of exactly these two pieces).  No CPU fallback: CPU tensors are moved to the device.
"""
from __future__ import annotations

import math
from typing import Sequence

import numpy as np
import torch

from . import ops
from ._lib import check, lib, ptr, stream


def _slaney_hz_to_mel(f):
    f = np.asarray(f, dtype=np.float64)
    return np.where(f >= 1000.0, 15.0 + np.log(np.maximum(f, 1e-30) / 1000.0) / (np.log(6.4) / 27.0), f * 3.0 / 200.0)


def _slaney_mel_to_hz(m):
    m = np.asarray(m, dtype=np.float64)
    return np.where(m >= 15.0, 1000.0 * np.exp((np.log(6.4) / 27.0) * (m - 15.0)), m * 200.0 / 3.0)


class MelFrontend:
    def __init__(self, device="cuda", sample_rate=16000, fft_size=1024, hop_size=256, num_mels=80, fmin=90.0,
                 fmax=7600.0, min_level_db=-100.0, ref_level_db=16.0):
        if fft_size % 4 or hop_size < 1 or hop_size > fft_size:
            raise ValueError("MelFrontend: fft_size must be a multiple of 4 and 1 <= hop_size <= fft_size")
        if not fmax <= sample_rate / 2:
            raise ValueError("MelFrontend: fmax above Nyquist")            # utils.py:115
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("MelFrontend runs on the HIP path only (no CPU fallback)")
        self.sr, self.fsize, self.hop, self.n_mels = sample_rate, fft_size, hop_size, num_mels
        self.min_level_db, self.ref_level_db = float(min_level_db), float(ref_level_db)
        self.min_level = float(np.exp(min_level_db / 20.0 * np.log(10.0)))  # utils.py:128
        nb = fft_size // 2 + 1
        self.nb, self.nbp = nb, (nb + 3) // 4 * 4
        n = np.arange(fft_size, dtype=np.float64)
        # lws "speech" analysis window: sqrt(periodic Hann * 2*hop/fsize)
        win = np.sqrt((0.5 - 0.5 * np.cos(2.0 * np.pi * n / fft_size)) * 2.0 * hop_size / fft_size)
        ang = 2.0 * np.pi * np.outer(np.arange(nb, dtype=np.float64), n) / fft_size
        basis = np.zeros((2 * self.nbp, fft_size), dtype=np.float64)
        basis[:nb] = np.cos(ang)
        basis[self.nbp:self.nbp + nb] = -np.sin(ang)
        # librosa.filters.mel (htk=False, Slaney area normalisation)
        fft_f = np.linspace(0.0, sample_rate / 2.0, nb)
        mel_f = _slaney_mel_to_hz(np.linspace(_slaney_hz_to_mel(fmin), _slaney_hz_to_mel(fmax), num_mels + 2))
        lower = (fft_f[None, :] - mel_f[:-2, None]) / (mel_f[1:-1] - mel_f[:-2])[:, None]
        upper = (mel_f[2:, None] - fft_f[None, :]) / (mel_f[2:] - mel_f[1:-1])[:, None]
        melw = np.zeros((num_mels, self.nbp), dtype=np.float64)
        melw[:, :nb] = np.maximum(0.0, np.minimum(lower, upper)) * (2.0 / (mel_f[2:] - mel_f[:-2]))[:, None]
        f32 = lambda a: torch.from_numpy(a.astype(np.float32)).to(self.device).contiguous()
        self.window, self.dft_basis, self.mel_basis = f32(win), f32(basis), f32(melw)
        self.mode = ops.MODE_F32

    # ---- utils.py:82-103
    def num_frames(self, length: int) -> int:
        pad = self.fsize - self.hop
        extra = 1 if length % self.hop == 0 else 2
        return (length + 2 * pad - self.fsize) // self.hop + extra

    def melspectrogram_batch(self, wavs: Sequence) -> list:
        """List of 1-D waveforms (numpy / tensors, any length >= 1) -> list of device tensors [80, M_i]."""
        L = lib()
        sigs = [torch.as_tensor(np.asarray(w, dtype=np.float32) if not torch.is_tensor(w) else w)
                .to(self.device, torch.float32).contiguous().view(-1) for w in wavs]
        if not sigs or any(s.numel() < 1 for s in sigs):
            raise ValueError("melspectrogram: empty waveform")
        ms = [self.num_frames(s.numel()) for s in sigs]
        rows = sum(ms)
        frames = torch.empty((rows, self.fsize), device=self.device, dtype=torch.float32)
        r = 0
        for s, m in zip(sigs, ms):
            check(L.dvae_stft_frames(ptr(s), s.numel(), ptr(self.window), frames[r:].data_ptr(), m, self.fsize,
                                     self.hop, self.fsize - self.hop, stream()), "dvae_stft_frames")
            r += m
        # features must not depend on the training compute mode (bf16 would put a leakage floor ~50 dB under each frame's
        # peak against a 100 dB normalisation range): both contractions are pinned to exact fp32 products
        reim = ops.linear_fwd(frames, self.dft_basis, None, mode=self.mode)  # [rows, 2*nbp]
        mag = torch.empty((rows, self.nbp), device=self.device, dtype=torch.float32)
        check(L.dvae_stft_magnitude(ptr(reim), ptr(mag), rows, self.nbp, stream()), "dvae_stft_magnitude")
        mel = ops.linear_fwd(mag, self.mel_basis, None, mode=self.mode)      # [rows, 80]
        outs, r = [], 0
        for m in ms:
            out = torch.empty((self.n_mels, m), device=self.device, dtype=torch.float32)
            check(L.dvae_mel_db_normalize(mel[r:].data_ptr(), ptr(out), m, self.n_mels, m, 0, self.min_level,
                                          self.ref_level_db, self.min_level_db, stream()), "dvae_mel_db_normalize")
            outs.append(out)
            r += m
        return outs

    def melspectrogram(self, wav):
        """One waveform -> [80, M] in [0, 1] (utils.py:68-73)."""
        return self.melspectrogram_batch([wav])[0]
