"""TEST INFRASTRUCTURE ONLY — deterministic, torch-RNG-independent weights and inputs.

Weights are 61-296 M floats, so fixtures never store them: the reference (in
tests/golden/make_golden.py), the oracle and the HIP model all regenerate the
same values from the `state_dict` key + shape and load them through
`load_state_dict` (SURVEY.md §8c "golden-vector recipe").  Ranges follow the
reference's initialisers (disentangled_vae.py:26-32: xavier-uniform Linear /
Conv1d; torch default U(+-1/sqrt(H)) for LSTM) but biases / BatchNorm affine
terms are made non-trivial so that tests exercise them.
"""
from __future__ import annotations

import math
import zlib
from typing import Dict

import numpy as np
import torch


def _rs(key: str, salt: int) -> np.random.RandomState:
    return np.random.RandomState((zlib.crc32(key.encode()) + 7919 * salt) % (2 ** 32))


def _uniform(key, salt, shape, lo, hi) -> torch.Tensor:
    a = _rs(key, salt).uniform(lo, hi, size=tuple(shape)).astype(np.float32)
    return torch.from_numpy(a)


def fill_state_dict(sd: Dict[str, torch.Tensor], salt: int = 0, random_running_stats: bool = False) -> Dict[str, torch.Tensor]:
    """Return a new state dict with the same keys/shapes/dtypes and deterministic values.
    `random_running_stats` gives BatchNorm non-trivial running statistics (for eval-mode tests)."""
    out = {}
    for k, v in sd.items():
        shape = tuple(v.shape)
        leaf = k.rsplit(".", 1)[-1]
        if leaf == "num_batches_tracked":
            out[k] = torch.zeros((), dtype=torch.long)
        elif leaf == "running_mean":
            out[k] = _uniform(k, salt, shape, -0.3, 0.3) if random_running_stats else torch.zeros(shape)
        elif leaf == "running_var":
            out[k] = _uniform(k, salt, shape, 0.5, 2.0) if random_running_stats else torch.ones(shape)
        elif "lstm" in k:
            hidden = shape[0] // 4
            b = 1.0 / math.sqrt(hidden)
            out[k] = _uniform(k, salt, shape, -b, b)
        elif v.dim() == 1:
            # BatchNorm affine (weight ~ 1, bias ~ 0) or a Linear/Conv bias
            is_bn_weight = leaf == "weight"
            if is_bn_weight:
                out[k] = _uniform(k, salt, shape, 0.5, 1.5)
            else:
                out[k] = _uniform(k, salt, shape, -0.1, 0.1)
        else:
            fan_out = shape[0] * int(np.prod(shape[2:])) if len(shape) > 2 else shape[0]
            fan_in = shape[1] * int(np.prod(shape[2:])) if len(shape) > 2 else shape[1]
            b = math.sqrt(6.0 / (fan_in + fan_out))
            out[k] = _uniform(k, salt, shape, -b, b)
        out[k] = out[k].to(v.dtype) if leaf != "num_batches_tracked" else out[k]
    return out


def synthetic_pair(batch: int, n_frames: int, seed: int = 1234):
    """x1, x2 ~ U[0,1) fp32 [B, 80, T] (mel range after preprocessing/encoder/utils.py:132-133)."""
    rs = np.random.RandomState(seed)
    x1 = rs.uniform(0.0, 1.0, size=(batch, 80, n_frames)).astype(np.float32)
    x2 = rs.uniform(0.0, 1.0, size=(batch, 80, n_frames)).astype(np.float32)
    return torch.from_numpy(x1), torch.from_numpy(x2)


def synthetic_eps(batch: int, speaker_size: int = 4, latent_dim: int = 32, seed: int = 99):
    """The three reparameterisation noises, in the reference's draw order."""
    rs = np.random.RandomState(seed)
    c = latent_dim - speaker_size
    mk = lambda d: torch.from_numpy(rs.standard_normal(size=(batch, d)).astype(np.float32))
    return mk(c), mk(c), mk(speaker_size)
