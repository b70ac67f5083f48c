"""Feeder of the training step: same-speaker utterance pairs of mel segments.

`SpeechDatasetGVAE` keeps the semantics of /root/reference/preprocessing/dataset.py:53-123:
directory layout `<root>/<speaker>/*.npy` with arrays [80, L]; per speaker the utterance list is
shuffled, cut into two halves and the i-th files of each half form a pair (:63-76, re-done by
`shuffle_data()` each epoch :78-91); each item is cropped to `samples_length` frames at a random
offset, or right-padded with zeros when shorter (:100-109); the label is the index of the speaker
directory (:111-112).  Differences: an explicit RNG (reproducible), sorted directory listing, and
L == samples_length crops at 0 instead of raising (np.random.choice(0) in the reference).

`SyntheticPairs` is the generator used by bench.py / tests: U[0,1) mels (the range produced by
preprocessing/encoder/utils.py:132-133) already resident on the device.
"""
from __future__ import annotations

import glob
import os
from typing import Optional

import numpy as np
import torch
from torch.utils.data import Dataset


class SpeechDatasetGVAE(Dataset):
    def __init__(self, file_path, sr=16000, samples_length=64, seed: Optional[int] = None):
        self.file_path = file_path
        self.sr = sr
        self.samples_length = samples_length
        self.rng = np.random.RandomState(seed) if seed is not None else np.random
        self.speaker_ids = sorted(d for d in os.listdir(file_path) if os.path.isdir(os.path.join(file_path, d)))
        self.spk_utt = [np.array(sorted(glob.glob(os.path.join(file_path, s, "*.npy")))) for s in self.speaker_ids]
        self.utterance_fp = np.empty((0, 2), dtype=object)
        self.shuffle_data()

    def shuffle_data(self):
        pairs = []
        for utt in self.spk_utt:
            self.rng.shuffle(utt)
            half = utt.shape[0] // 2
            pairs += [(utt[i], utt[half + i]) for i in range(half)]
        self.utterance_fp = np.array(pairs, dtype=object).reshape(-1, 2)

    def _crop(self, mel):
        T = self.samples_length
        L = mel.shape[1]
        if L < T:
            return np.pad(mel, ((0, 0), (0, T - L)), "constant", constant_values=0)
        start = int(self.rng.randint(0, L - T)) if L > T else 0
        return mel[:, start:start + T]

    def __getitem__(self, index):
        u1, u2 = self.utterance_fp[index]
        mel1, mel2 = self._crop(np.load(u1)), self._crop(np.load(u2))
        spk = self.speaker_ids.index(os.path.basename(os.path.dirname(u1)))
        return torch.from_numpy(np.ascontiguousarray(mel1)), torch.from_numpy(np.ascontiguousarray(mel2)), \
            torch.tensor(spk)

    def __len__(self):
        return len(self.utterance_fp)

    def get_utterance(self, speaker, utterance):
        return np.load(os.path.join(self.file_path, speaker, utterance))


def write_synthetic_corpus(root, n_speakers=2, n_utt=8, n_mel=80, length=96, seed=0):
    """BASELINE config[0]: `n_speakers` x `n_utt` float64 .npy files of shape [80, length], U[0,1)."""
    rs = np.random.RandomState(seed)
    for s in range(n_speakers):
        d = os.path.join(root, f"spk{s:03d}")
        os.makedirs(d, exist_ok=True)
        for u in range(n_utt):
            np.save(os.path.join(d, f"utt{u:03d}_mel.npy"), rs.uniform(0.0, 1.0, size=(n_mel, length)))
    return root


class SyntheticPairs:
    """Device-resident synthetic batches: x1, x2 ~ U[0,1) fp32 [B, 80, T]; speaker ids uniform over n_speakers
    (ids are unused by the loss, variational_base_vae.py:58)."""

    def __init__(self, batch, n_frames, n_speakers=10, seed=1234, device="cuda", n_mel=80):
        rs = np.random.RandomState(seed)
        self.x1 = torch.from_numpy(rs.uniform(0, 1, size=(batch, n_mel, n_frames)).astype(np.float32)).to(device)
        self.x2 = torch.from_numpy(rs.uniform(0, 1, size=(batch, n_mel, n_frames)).astype(np.float32)).to(device)
        self.spk = torch.from_numpy(rs.randint(0, n_speakers, size=(batch,))).to(device)

    def batch(self):
        return self.x1, self.x2, self.spk


class GpuPairLoader:
    """Device-resident replacement for `DataLoader(SpeechDatasetGVAE, shuffle=True)` (train.py:55-56).

    With a ~36 ms step the reference's loader (num_workers=0, two np.load + a host->device copy per item) would
    dominate, so the whole corpus is uploaded ONCE as a padded [n_utt, 80, Lmax] fp32 tensor (VCTK mels are a few
    GB: trivial against 288 GB of HBM) and a batch is produced by one HIP gather kernel.  Pairing, per-epoch
    re-pairing (`dataset.shuffle_data()`), random crop / right zero-pad and the speaker label are those of
    SpeechDatasetGVAE — the pair list IS the dataset's; only where the bytes live and who crops changes.
    Yields (x1 [B,80,T], x2 [B,80,T], speaker_ids [B]) on the device.

    Data parallel: with `world_size` > 1 every rank holds the whole corpus (it is small) and takes pairs
    rank, rank+world, ... of the SAME epoch permutation, so `seed` (and the dataset's seed) must be given and equal
    on all ranks; every rank sees the same number of batches (the remainder is dropped)."""

    def __init__(self, dataset: SpeechDatasetGVAE, batch_size: int, device="cuda", seed: Optional[int] = None,
                 drop_last: bool = True, shuffle: bool = True, rank: int = 0, world_size: int = 1):
        if world_size > 1 and seed is None:
            raise ValueError("GpuPairLoader: data-parallel sharding needs an explicit seed shared by all ranks")
        if not 0 <= rank < world_size:
            raise ValueError("GpuPairLoader: rank out of range")
        self.dataset = dataset
        self.batch_size, self.drop_last, self.shuffle = batch_size, drop_last, shuffle
        self.rank, self.world_size = rank, world_size
        self.rng = np.random.RandomState(seed) if seed is not None else np.random
        self.crop_rng = np.random.RandomState(seed + 7919 * (rank + 1)) if seed is not None else np.random
        self.T = dataset.samples_length
        files = sorted({f for utts in dataset.spk_utt for f in utts})
        self.index = {f: i for i, f in enumerate(files)}
        arrs = [np.load(f) for f in files]
        self.C = arrs[0].shape[0]
        self.lengths = np.array([a.shape[1] for a in arrs], dtype=np.int32)
        self.Lmax = int(self.lengths.max())
        host = np.zeros((len(arrs), self.C, self.Lmax), dtype=np.float32)
        for i, a in enumerate(arrs):
            host[i, :, :a.shape[1]] = a
        self.device = torch.device(device)
        self.mels = torch.from_numpy(host).to(self.device)
        self.lens_dev = torch.from_numpy(self.lengths).to(self.device)
        self.last_meta = None       # (utt1, utt2, off1, off2) of the last batch, for tests

    def _n_local(self):
        return len(self.dataset) // self.world_size

    def __len__(self):
        n = self._n_local()
        return n // self.batch_size if self.drop_last else (n + self.batch_size - 1) // self.batch_size

    def _offset(self, L):
        # same rule as SpeechDatasetGVAE._crop: random start if longer, 0 (and zero padding) otherwise
        return int(self.crop_rng.randint(0, L - self.T)) if L > self.T else 0

    def gather(self, utt, off):
        """Crop batch on the device: utt, off int arrays [n] -> [n, 80, T]."""
        from ._lib import check, lib, ptr, stream
        n = len(utt)
        u = torch.as_tensor(np.asarray(utt, dtype=np.int32)).to(self.device)
        o = torch.as_tensor(np.asarray(off, dtype=np.int32)).to(self.device)
        out = torch.empty((n, self.C, self.T), device=self.device, dtype=torch.float32)
        check(lib().dvae_gather_crop(ptr(self.mels), ptr(self.lens_dev), ptr(u), ptr(o), ptr(out), n, self.C, self.T,
                                     self.Lmax, stream()), "dvae_gather_crop")
        return out

    def sample_batch(self, seed: int = 0):
        """One batch drawn with a PRIVATE generator: leaves the epoch permutation and crop streams untouched, so a rank
        that looks at a batch on its own (rank 0's estimate_trained_model) stays in step with its peers."""
        rs = np.random.RandomState(seed)
        n = len(self.dataset)
        sel = rs.permutation(n)[:min(self.batch_size, n)]
        pairs = self.dataset.utterance_fp[sel]
        u1 = np.array([self.index[p[0]] for p in pairs], dtype=np.int32)
        u2 = np.array([self.index[p[1]] for p in pairs], dtype=np.int32)
        off = lambda us: np.array([int(rs.randint(0, int(self.lengths[u]) - self.T)) if self.lengths[u] > self.T else 0
                                   for u in us], dtype=np.int32)
        spk = np.array([self.dataset.speaker_ids.index(os.path.basename(os.path.dirname(p[0]))) for p in pairs])
        return self.gather(u1, off(u1)), self.gather(u2, off(u2)), torch.from_numpy(spk).to(self.device)

    def __iter__(self):
        n = len(self.dataset)
        order = self.rng.permutation(n) if self.shuffle else np.arange(n)     # identical on every rank
        order = order[:self._n_local() * self.world_size][self.rank::self.world_size]
        for b in range(len(self)):
            sel = order[b * self.batch_size:(b + 1) * self.batch_size]
            pairs = self.dataset.utterance_fp[sel]
            u1 = np.array([self.index[p[0]] for p in pairs], dtype=np.int32)
            u2 = np.array([self.index[p[1]] for p in pairs], dtype=np.int32)
            o1 = np.array([self._offset(int(self.lengths[u])) for u in u1], dtype=np.int32)
            o2 = np.array([self._offset(int(self.lengths[u])) for u in u2], dtype=np.int32)
            spk = np.array([self.dataset.speaker_ids.index(os.path.basename(os.path.dirname(p[0]))) for p in pairs])
            self.last_meta = (u1, u2, o1, o2)
            yield self.gather(u1, o1), self.gather(u2, o2), torch.from_numpy(spk).to(self.device)
