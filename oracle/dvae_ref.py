"""TEST INFRASTRUCTURE ONLY — CPU oracle for the disentangled-VAE training step.

A plain-PyTorch (fp32, device-agnostic) restatement of the reference hot path:

  * network        /root/reference/model/disentangled_vae.py:43-87 (Postnet),
                   :124-195 (layer table), :198-220 (encode), :222-228
                   (reparameterise), :230-248 (decode), :250-279 (pair forward)
  * loss           /root/reference/model/disentangled_vae.py:310-327
  * train step     /root/reference/model/variational_base_vae.py:58-70
  * optimiser      torch.optim.Adam(lr) created at disentangled_vae.py:304

Differences from the reference, all deliberate and all inert for parity:
  - the two frame-count dependent Linear layers are sized `n_frames*2*dim_neck`
    instead of the literal 8192 (= 64*128) so T != 64 works;
  - the three reparameterisation noises are passed in explicitly (the reference
    draws them from the CPU global generator, disentangled_vae.py:224);
  - no `.cuda()` calls; runs wherever its parameters live.

The module tree reproduces the reference's `state_dict` keys exactly, so one
deterministic weight set (oracle/fill.py) loads into the reference, this
oracle and the HIP model alike.

Pinned by tests/test_oracle_golden.py against vectors produced by running the
imported reference itself (tests/golden/make_golden.py).
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Sequence, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

N_MEL = 80
CONV_CH = 512
KSIZE = 5


class _Keyed(nn.Module):
    """One child stored under a chosen attribute name (gives the `.conv.` and
    `.linear_layer.` key levels of ConvNorm / LinearNorm, disentangled_vae.py:90-121)."""

    def __init__(self, attr: str, child: nn.Module):
        super().__init__()
        self._attr = attr
        setattr(self, attr, child)

    def forward(self, x):
        return getattr(self, self._attr)(x)


def _conv(cin: int, cout: int, keyed: bool) -> nn.Module:
    c = nn.Conv1d(cin, cout, KSIZE, stride=1, padding=KSIZE // 2)
    return _Keyed("conv", c) if keyed else c


def _conv_bn(cin: int, cout: int, keyed: bool) -> nn.Sequential:
    return nn.Sequential(_conv(cin, cout, keyed), nn.BatchNorm1d(cout))


def _linear(i: int, o: int, keyed: bool) -> nn.Module:
    lin = nn.Linear(i, o)
    return _Keyed("linear_layer", lin) if keyed else lin


class _PostnetRef(nn.Module):
    # reference: Postnet, disentangled_vae.py:43-87
    def __init__(self):
        super().__init__()
        chans = [N_MEL] + [CONV_CH] * 4 + [N_MEL]
        self.convolutions = nn.ModuleList(
            _conv_bn(chans[i], chans[i + 1], keyed=True) for i in range(5))

    def forward(self, x):
        for blk in self.convolutions[:-1]:
            x = torch.tanh(blk(x))
        return self.convolutions[-1](x)


class RefDVAE(nn.Module):
    """Restatement of DisentangledVAE (disentangled_vae.py:124-279)."""

    def __init__(self, speaker_size: int = 4, latent_dim: int = 32, n_frames: int = 64,
                 dim_neck: int = 64, dim_pre: int = 512):
        super().__init__()
        self.speaker_size = speaker_size
        self.latent_dim = latent_dim
        self.dim_neck = dim_neck
        self.n_frames = n_frames
        flat = n_frames * 2 * dim_neck

        self.postnet = _PostnetRef()
        self.enc_modules = nn.ModuleList(
            _conv_bn(N_MEL if i == 0 else CONV_CH, CONV_CH, keyed=True) for i in range(3))
        self.enc_lstm = nn.LSTM(dim_pre, dim_neck, 2, batch_first=True, bidirectional=True)
        self.enc_linear = _linear(flat, 2048, keyed=True)
        self.style = _linear(2048, 2 * speaker_size, keyed=True)
        self.content = _linear(2048, 2 * (latent_dim - speaker_size), keyed=True)

        self.dec_pre_linear1 = _linear(latent_dim, 2048, keyed=False)
        self.dec_pre_linear2 = _linear(2048, flat, keyed=False)
        self.dec_lstm1 = nn.LSTM(2 * dim_neck, 512, 1, batch_first=True)
        self.dec_modules = nn.ModuleList(_conv_bn(dim_pre, dim_pre, keyed=False) for _ in range(3))
        self.dec_lstm2 = nn.LSTM(dim_pre, 1024, 2, batch_first=True)
        self.dec_linear2 = _linear(1024, N_MEL, keyed=True)
        self.reset_like_reference()

    # reference: init_weights applied to every Linear / Conv1d (disentangled_vae.py:26-32,195)
    def reset_like_reference(self):
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.xavier_uniform_(m.weight)
                m.bias.data.fill_(0.01)
            elif isinstance(m, nn.Conv1d):
                nn.init.xavier_uniform_(m.weight)
                m.bias.data.zero_()

    # disentangled_vae.py:198-220
    def encode(self, x):
        nb = x.shape[0]
        for blk in self.enc_modules:
            x = F.relu(blk(x))
        seq, _ = self.enc_lstm(x.transpose(1, 2))
        feat = F.relu(self.enc_linear(seq.reshape(nb, -1)))
        st, ct = self.style(feat), self.content(feat)
        s, c = self.speaker_size, self.latent_dim - self.speaker_size
        return st[:, :s], st[:, s:], ct[:, :c], ct[:, c:]

    # disentangled_vae.py:222-228 with the noise made explicit
    @staticmethod
    def reparameterize(mu, logvar, eps):
        if eps is None:
            return mu
        return eps * torch.exp(0.5 * logvar) + mu

    # disentangled_vae.py:230-248
    def decode(self, z):
        h = self.dec_pre_linear2(self.dec_pre_linear1(z))
        h = h.view(z.shape[0], -1, 2 * self.dim_neck)
        h, _ = self.dec_lstm1(h)
        h = h.transpose(1, 2)
        for blk in self.dec_modules:
            h = F.relu(blk(h))
        h, _ = self.dec_lstm2(h.transpose(1, 2))
        return self.dec_linear2(h).transpose(1, 2)

    # disentangled_vae.py:250-279; eps = (eps_content1, eps_content2, eps_style)
    def forward(self, x1, x2, eps: Sequence[Optional[torch.Tensor]]):
        e_c1, e_c2, e_s = eps
        s_mu1, s_lv1, c_mu1, c_lv1 = self.encode(x1)
        z_c1 = self.reparameterize(c_mu1, c_lv1, e_c1)
        s_mu2, s_lv2, c_mu2, c_lv2 = self.encode(x2)
        z_c2 = self.reparameterize(c_mu2, c_lv2, e_c2)
        s_mu = (s_mu1 + s_mu2.detach()) / 2
        s_lv = (s_lv1 + s_lv2.detach()) / 2
        z_s = self.reparameterize(s_mu, s_lv, e_s)
        q1_mu, q1_lv = torch.cat((s_mu, c_mu1), -1), torch.cat((s_lv, c_lv1), -1)
        q2_mu, q2_lv = torch.cat((s_mu, c_mu2), -1), torch.cat((s_lv, c_lv2), -1)
        r1 = self.decode(torch.cat((z_s, z_c1), -1))
        r2 = self.decode(torch.cat((z_s, z_c2), -1))
        r1_hat = r1 + self.postnet(r1)
        r2_hat = r2 + self.postnet(r2)
        return r1, r2, r1_hat, r2_hat, q1_mu, q1_lv, q2_mu, q2_lv, s_mu, s_lv


def _kl_terms(mu, lv):
    return 1 + lv - mu.pow(2) - lv.exp()


def loss_gvae2(x1, x2, outs, batch_size: int, mse_cof: float = 10.0, kl_cof: float = 10.0):
    """Restatement of loss_functionGVAE2 (disentangled_vae.py:310-327).
    The L1 sums are divided by the CONFIGURED batch size, not the actual one."""
    r1, r2, r1_hat, r2_hat, q1_mu, q1_lv, q2_mu, q2_lv, s_mu, s_lv = outs
    l1 = lambda a, b: (a - b).abs().sum() / batch_size
    rec = [l1(x1, r1), l1(x2, r2), l1(x1, r1_hat), l1(x2, r2_hat)]
    kl1 = -0.5 * _kl_terms(q1_mu, q1_lv).sum(-1).mean()
    kl2 = -0.5 * _kl_terms(q2_mu, q2_lv).sum(-1).mean()
    kl_style = -_kl_terms(s_mu, s_lv).sum() / batch_size
    total = mse_cof * (rec[0] + rec[1] + rec[2] + rec[3]) + kl_cof * (kl1 + kl2)
    return (total, rec[0], rec[1], rec[2], rec[3], kl1, kl2, kl_style)


class RefTrainer:
    """Restatement of ConvolutionalMulVAE + VariationalBaseModelVAE.step
    (disentangled_vae.py:288-304, variational_base_vae.py:58-70)."""

    def __init__(self, batch_size: int, speaker_size: int = 4, latent_dim: int = 32,
                 n_frames: int = 64, lr: float = 1e-4, mse_cof: float = 10.0,
                 kl_cof: float = 10.0, device="cpu"):
        self.batch_size = batch_size
        self.mse_cof, self.kl_cof, self.lr = mse_cof, kl_cof, lr
        self.model = RefDVAE(speaker_size, latent_dim, n_frames).to(device)
        self.optimizer = torch.optim.Adam(self.model.parameters(), lr=lr)

    def draw_eps(self, nb: int, generator: Optional[torch.Generator] = None):
        """Same draw order and shapes as the reference's three normal_() calls
        (disentangled_vae.py:252,255,261)."""
        c = self.model.latent_dim - self.model.speaker_size
        s = self.model.speaker_size
        mk = lambda d: torch.empty(nb, d).normal_(generator=generator)
        return mk(c), mk(c), mk(s)

    def step(self, x1, x2, eps, train: bool = True):
        if train:
            self.optimizer.zero_grad()
        outs = self.model(x1, x2, eps)
        losses = loss_gvae2(x1, x2, outs, self.batch_size, self.mse_cof, self.kl_cof)
        if train:
            losses[0].backward()
            self.optimizer.step()
        return tuple(float(v.item()) for v in losses)


def chunked_step_grads(tr: RefTrainer, x1, x2, eps, n_chunks: int):
    """Gradient of a data-parallel step without communication: split the batch
    into `n_chunks` rank-local shards (BatchNorm statistics per shard, loss
    divided by the shard size), average the shard gradients.  This is what an
    N-rank step must equal.  Returns {name: grad}."""
    nb = x1.shape[0]
    assert nb % n_chunks == 0
    per = nb // n_chunks
    acc: Dict[str, torch.Tensor] = {}
    for r in range(n_chunks):
        sl = slice(r * per, (r + 1) * per)
        tr.model.zero_grad(set_to_none=True)
        outs = tr.model(x1[sl], x2[sl], tuple(e[sl] for e in eps))
        loss = loss_gvae2(x1[sl], x2[sl], outs, per, tr.mse_cof, tr.kl_cof)[0]
        loss.backward()
        for n, p in tr.model.named_parameters():
            g = p.grad.detach().clone() / n_chunks
            acc[n] = g if n not in acc else acc[n] + g
    return acc


# ------------------------------------------------------------------------------------------------
# Mel -> mel conversion (inference), tensor part of voice_conversion_mel
# (/root/reference/model/variational_base_vae.py:269-298) and chunking_mel (:335-348).
def chunk_mel(mel: torch.Tensor, n_frames: int = 64) -> torch.Tensor:
    """[80, L] -> [L//T + 1, 80, T]; the last chunk is right-padded with zeros (when L % T == 0 that is one
    extra all-zero chunk, as in the reference)."""
    n = mel.shape[1] // n_frames + 1
    out = torch.zeros((n, mel.shape[0], n_frames), dtype=mel.dtype)
    for i in range(n):
        part = mel[:, i * n_frames:(i + 1) * n_frames]
        out[i, :, :part.shape[1]] = part
    return out


@torch.no_grad()
def convert_mel_ref(model: "RefDVAE", source_mel: torch.Tensor, target_mel: torch.Tensor):
    """Eval-mode conversion of one utterance: content of the source, style (mean style_mu over chunks) of the target.
    Returns dict(source, recons, converted, spectral_detail), each [80, n_chunks*T] (converted clamped to [0,1])."""
    model.eval()
    T = model.n_frames
    src, trg = chunk_mel(source_mel.float(), T), chunk_mel(target_mel.float(), T)
    s_mu, _, c_mu, _ = model.encode(src)
    t_mu, _, _, _ = model.encode(trg)
    n = src.shape[0]
    src_style = s_mu.mean(0, keepdim=True).repeat(n, 1)
    trg_style = t_mu.mean(0, keepdim=True).repeat(n, 1)
    recons = model.decode(torch.cat((src_style, c_mu), -1))
    conv = model.decode(torch.cat((trg_style, c_mu), -1))
    conv = conv + model.postnet(conv)
    cat = lambda x: torch.cat([x[i] for i in range(x.shape[0])], 1)
    recons, conv, source = cat(recons), torch.clamp(cat(conv), 0.0, 1.0), cat(src)
    return {"source": source, "recons": recons, "converted": conv, "spectral_detail": source * (recons / conv)}
