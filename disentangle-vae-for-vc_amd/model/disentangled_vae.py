"""MI355X-native DisentangledVAE / ConvolutionalMulVAE.

Drop-in for /root/reference/model/disentangled_vae.py: same class names, constructor arguments,
method signatures, return tuples and `state_dict` keys — but `torch.nn` is used
only as a PARAMETER CONTAINER.  No torch.nn forward is ever called: all tensor math goes through
the hand-written HIP kernels of libdvae_hip.so (ops.py), forward and backward.

MI355X-first design decisions (none of them visible through the API):
  * the utterance pair (x1, x2) runs through encoder, decoder and postnet ONCE as a batch of
    N = 2B mel segments (x1 rows then x2 rows); BatchNorm statistics stay per utterance of the
    pair (2 groups), exactly as the reference's two separate calls produce them;
  * activations are frame-major [T, N, C]: a conv tap is a row shift, an LSTM frame is one
    contiguous slab, every GEMM-shaped op is one MFMA contraction over rows;
  * parameters/gradients/moments live in flat buffers (optim.FlatAdam);
  * Conv1d weights LIVE in the packed layout [5][Cout][Cin] the implicit-GEMM kernels read (no forward pack, the
    weight gradient accumulates straight into the flat gradient buffer); `state_dict()` / `load_state_dict()` convert
    to / from torch's [Cout][Cin][5], so checkpoints stay interchangeable with the reference;
  * every other weight-derived operand layout (transposed conv packs, W_ih^T, W_hh^T, LSTM fragment packs,
    b_ih + b_hh) is refreshed by ONE launch at the start of a forward (derived.DerivedWeights).

Extra constructor argument `n_frames` (default 64) sizes the two layers the reference hard-codes
to 8192 = 64*128 (disentangled_vae.py:165,171) so T = 128/256/512 work.
"""
from __future__ import annotations

import math
from typing import Optional, Sequence

import torch
import torch.nn as nn

from .. import ops
from ..ops import (ACT_NONE, fanout, ACT_RELU, ACT_TANH, ConvBnActFn, FramesToMelFn, LatentFn, LinearFn, LstmLayerFn,
                   LstmStack2Fn, Permute102Fn, mel_to_frames)
from ..derived import DerivedWeights
from ..optim import FlatAdam
from .variational_base_vae import VariationalBaseModelVAE

N_MEL = 80


# ------------------------------------------------------------------ parameter containers
class _Conv1dParams(nn.Module):
    """Conv1d(k=5) parameters.  `weight` is stored PACKED, [5][Cout][Cin] (one [Cout][Cin] matrix per tap: the B
    operand of the implicit-GEMM kernel); `state_dict()` emits and `load_state_dict()` accepts torch's [Cout][Cin][5]
    (the reference's checkpoint layout) through the two hooks below."""

    def __init__(self, cin, cout, k=5):
        super().__init__()
        if k != 5:
            raise ValueError("the HIP conv kernels are built for kernel_size 5 (every conv of the reference)")
        self.cin, self.cout, self.k = cin, cout, k
        self.weight = nn.Parameter(torch.empty(k, cout, cin))
        self.bias = nn.Parameter(torch.zeros(cout))
        self.reset()
        self._register_state_dict_hook(self._emit_torch_layout)
        self._register_load_state_dict_pre_hook(self._accept_torch_layout)

    def reset(self):
        """xavier-uniform over the torch layout's fans (fan_in = Cin*5, fan_out = Cout*5), bias 0 (init_weights)."""
        w = torch.empty(self.cout, self.cin, self.k)
        nn.init.xavier_uniform_(w)
        with torch.no_grad():
            self.weight.copy_(w.permute(2, 0, 1))
            self.bias.fill_(0)

    @staticmethod
    def _emit_torch_layout(module, state_dict, prefix, local_metadata):
        state_dict[prefix + "weight"] = state_dict[prefix + "weight"].permute(1, 2, 0).contiguous()

    def _accept_torch_layout(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        k = prefix + "weight"
        if k in state_dict and tuple(state_dict[k].shape) == (self.cout, self.cin, self.k):
            state_dict[k] = state_dict[k].permute(2, 0, 1).contiguous()

    def torch_layout(self, t):
        """A tensor shaped like `weight` (the weight itself, its gradient, an Adam moment) as [Cout][Cin][5]."""
        return t.permute(1, 2, 0)


class ConvNorm(nn.Module):
    """Key level `.conv.` of the reference's ConvNorm (disentangled_vae.py:103-121)."""

    def __init__(self, in_channels, out_channels, kernel_size=5, **_):
        super().__init__()
        self.conv = _Conv1dParams(in_channels, out_channels, kernel_size)


class _BatchNormParams(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(c))
        self.bias = nn.Parameter(torch.zeros(c))
        self.register_buffer("running_mean", torch.zeros(c))
        self.register_buffer("running_var", torch.ones(c))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))


class _LinearParams(nn.Module):
    def __init__(self, i, o):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(o, i))
        self.bias = nn.Parameter(torch.full((o,), 0.01))
        nn.init.xavier_uniform_(self.weight)


class LinearNorm(nn.Module):
    """Key level `.linear_layer.` of the reference's LinearNorm (disentangled_vae.py:90-100)."""

    def __init__(self, in_dim, out_dim, **_):
        super().__init__()
        self.linear_layer = _LinearParams(in_dim, out_dim)


class _LSTMParams(nn.Module):
    """Parameters of nn.LSTM with torch's names/layout/default init U(+-1/sqrt(H)); gate order i,f,g,o."""

    def __init__(self, input_size, hidden_size, num_layers=1, bidirectional=False):
        super().__init__()
        self.input_size, self.hidden_size = input_size, hidden_size
        self.num_layers, self.bidirectional = num_layers, bidirectional
        b = 1.0 / math.sqrt(hidden_size)
        for l in range(num_layers):
            in_l = input_size if l == 0 else hidden_size * (2 if bidirectional else 1)
            for sfx in ([""] + (["_reverse"] if bidirectional else [])):
                for name, shape in (("weight_ih", (4 * hidden_size, in_l)), ("weight_hh", (4 * hidden_size, hidden_size)),
                                    ("bias_ih", (4 * hidden_size,)), ("bias_hh", (4 * hidden_size,))):
                    p = nn.Parameter(torch.empty(*shape).uniform_(-b, b))
                    setattr(self, f"{name}_l{l}{sfx}", p)

    def layer(self, l):
        g = lambda n: getattr(self, n)
        fwd = (g(f"weight_ih_l{l}"), g(f"weight_hh_l{l}"), g(f"bias_ih_l{l}"), g(f"bias_hh_l{l}"))
        if self.bidirectional:
            return fwd + (g(f"weight_ih_l{l}_reverse"), g(f"weight_hh_l{l}_reverse"), g(f"bias_ih_l{l}_reverse"),
                          g(f"bias_hh_l{l}_reverse"))
        return fwd + (None, None, None, None)


def _conv_bn(cin, cout, keyed):
    return nn.Sequential(ConvNorm(cin, cout) if keyed else _Conv1dParams(cin, cout), _BatchNormParams(cout))


def _conv_of(block):
    c = block[0]
    return c.conv if isinstance(c, ConvNorm) else c


def init_weights(m):
    """reference init_weights (disentangled_vae.py:26-32): xavier-uniform Linear (bias 0.01) and Conv1d (bias 0)."""
    if isinstance(m, _LinearParams):
        nn.init.xavier_uniform_(m.weight)
        m.bias.data.fill_(0.01)
    if isinstance(m, _Conv1dParams):
        m.reset()


class Postnet(nn.Module):
    """Five conv(k5)+BatchNorm blocks, tanh on all but the last (disentangled_vae.py:43-87)."""

    def __init__(self):
        super().__init__()
        ch = [N_MEL, 512, 512, 512, 512, N_MEL]
        self.convolutions = nn.ModuleList(_conv_bn(ch[i], ch[i + 1], keyed=True) for i in range(5))

    def forward_frames(self, y, n_seg, groups, residual=None, wpt=None, w16=None):
        """y [T*N, 80] frame-major -> postnet(y) (+ residual fused into the last BatchNorm apply).
        `wpt`: the five transposed conv packs from the owning model's DerivedWeights (None: made in backward)."""
        last = len(self.convolutions) - 1
        y16 = None            # bf16 compute mode: the data of y when y is a placeholder (ops.ConvBnActFn)
        for i, blk in enumerate(self.convolutions):
            c, bn = _conv_of(blk), blk[1]
            out = ConvBnActFn.apply(y, c.weight, c.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var,
                                    bn.num_batches_tracked, residual if i == last else None, n_seg, groups,
                                    ACT_TANH if i < last else ACT_NONE, self.training,
                                    None if wpt is None else wpt[i], None if w16 is None else w16[i], y16,
                                    i < last)          # the last block's output leaves the conv stack: real fp32
            y, y16 = out if i < last else (out, None)
        return y

    def forward(self, x):
        """x [B, 80, T] -> [B, 80, T], as called by voice conversion (variational_base_vae.py:292)."""
        B, C, T = x.shape
        y = self.forward_frames(mel_to_frames(x.contiguous()), B, 1)
        return FramesToMelFn.apply(y, B, C, T)

    def forward_plus_input(self, x):
        """x + postnet(x) with the residual fused into the last BatchNorm apply (variational_base_vae.py:292-293)."""
        B, C, T = x.shape
        xf = mel_to_frames(x.contiguous())
        return FramesToMelFn.apply(self.forward_frames(xf, B, 1, residual=xf), B, C, T)


class DisentangledVAE(nn.Module):
    """reference: disentangled_vae.py:124-286."""

    def __init__(self, speaker_size, input_sz=(1, 64, 80), kernel_szs=[512, 512, 512], hidden_sz: int = 256,
                 latent_sz: int = 32, c: float = 512, c_delta: float = 0.001, beta: float = 0.1,
                 beta_delta: float = 0, dim_neck=64, latent_dim=64, dim_pre=512, batch_size=10, n_frames: int = 64):
        super().__init__()
        # the contraction kernels move operands in 16-byte pieces: the two Gaussian heads (mu | logvar) need an even
        # width each and the latent (decoder input) a multiple of 4 — the reference accepts any size, this path says so
        # here instead of failing inside the first backward
        c_sz = latent_dim - speaker_size
        if speaker_size < 1 or c_sz < 1 or speaker_size % 2 or c_sz % 2 or latent_dim % 4:
            raise ValueError(f"DisentangledVAE(HIP): speaker_size={speaker_size} and latent_dim-speaker_size={c_sz} must "
                             f"be even and latent_dim={latent_dim} a multiple of 4 (16-byte operand rows)")
        if dim_neck != 64 or dim_pre != 512:
            raise ValueError("DisentangledVAE(HIP): dim_neck=64 and dim_pre=512 are the sizes the reference hard-codes "
                             "(disentangled_vae.py:165,171: 8192 = 64 frames x 2*dim_neck) and the ones built here")
        self.batch_size = batch_size
        self._input_sz = input_sz
        self._channel_szs = [input_sz[0]] + list(kernel_szs)
        self._hidden_sz = hidden_sz
        self._c, self._c_delta = c, c_delta
        self._beta, self._beta_delta = beta, beta_delta
        self.latent_dim = latent_dim
        self.dim_neck = dim_neck
        self.speaker_size = speaker_size
        self.n_frames = n_frames
        flat = n_frames * 2 * dim_neck

        self.postnet = Postnet()
        self.enc_modules = nn.ModuleList(_conv_bn(N_MEL if i == 0 else 512, 512, keyed=True) for i in range(3))
        self.enc_lstm = _LSTMParams(dim_pre, dim_neck, 2, bidirectional=True)
        self.enc_linear = LinearNorm(flat, 2048)
        self.style = LinearNorm(2048, speaker_size * 2)
        self.content = LinearNorm(2048, (latent_dim - speaker_size) * 2)

        self.dec_pre_linear1 = _LinearParams(latent_dim, 2048)
        self.dec_pre_linear2 = _LinearParams(2048, flat)
        self.dec_lstm1 = _LSTMParams(dim_neck * 2, 512, 1)
        self.dec_modules = nn.ModuleList(_conv_bn(dim_pre, dim_pre, keyed=False) for _ in range(3))
        self.dec_lstm2 = _LSTMParams(dim_pre, 1024, 2)
        self.dec_linear2 = LinearNorm(1024, N_MEL)
        self.apply(init_weights)
        self._derived: Optional[DerivedWeights] = None

        # explicit reparameterisation noise for parity runs: (eps_content1, eps_content2, eps_style),
        # the reference's three normal_() draws in call order (disentangled_vae.py:252,255,261)
        self.eps_override: Optional[Sequence[torch.Tensor]] = None

    # ---- parameters in the order their gradients complete during backward (for bucketed all-reduce)
    def backward_param_order(self):
        order = []
        take = lambda mod, pre: order.extend((pre + n, p) for n, p in mod.named_parameters())
        for i in reversed(range(5)):
            take(self.postnet.convolutions[i], f"postnet.convolutions.{i}.")
        take(self.dec_linear2, "dec_linear2.")
        take(self.dec_lstm2, "dec_lstm2.")
        for i in reversed(range(3)):
            take(self.dec_modules[i], f"dec_modules.{i}.")
        take(self.dec_lstm1, "dec_lstm1.")
        take(self.dec_pre_linear2, "dec_pre_linear2.")
        take(self.dec_pre_linear1, "dec_pre_linear1.")
        take(self.content, "content.")
        take(self.style, "style.")
        take(self.enc_linear, "enc_linear.")
        take(self.enc_lstm, "enc_lstm.")
        for i in reversed(range(3)):
            take(self.enc_modules[i], f"enc_modules.{i}.")
        assert len(order) == len(list(self.parameters()))
        return order

    # ---- weight-derived operand layouts (one launch per forward, see derived.py)
    def _lstm_names(self):
        for mname in ("enc_lstm", "dec_lstm1", "dec_lstm2"):
            mod = getattr(self, mname)
            for l in range(mod.num_layers):
                ps = mod.layer(l)
                yield f"{mname}.{l}.0", ps[:4]
                if mod.bidirectional:
                    yield f"{mname}.{l}.1", ps[4:]

    def _refresh_derived(self):
        if self._derived is None:
            convs = {}
            for i in range(1, 3):                                   # enc_modules[0] reads the mel: no data gradient
                convs[f"enc_modules.{i}"] = _conv_of(self.enc_modules[i]).weight
            for i in range(3):
                convs[f"dec_modules.{i}"] = _conv_of(self.dec_modules[i]).weight
            for i in range(5):
                convs[f"postnet.{i}"] = _conv_of(self.postnet.convolutions[i]).weight
            casts = {f"enc_modules.{i}": _conv_of(self.enc_modules[i]).weight for i in range(3)}
            casts.update({f"dec_modules.{i}": _conv_of(self.dec_modules[i]).weight for i in range(3)})
            casts.update({f"postnet.{i}": _conv_of(self.postnet.convolutions[i]).weight for i in range(5)})
            for n in ("enc_linear", "style", "content", "dec_linear2"):
                casts[n] = getattr(self, n).linear_layer.weight
            casts["dec_pre_linear1"], casts["dec_pre_linear2"] = self.dec_pre_linear1.weight, self.dec_pre_linear2.weight
            self._derived = DerivedWeights(convs, dict(self._lstm_names()), casts)
        self._derived.refresh(ops.current_mode())
        return self._derived

    def _wpt(self, name):
        return self._derived.wpt.get(name) if self._derived is not None else None

    def _w16(self, name):
        """bf16 copy of a weight (bf16 compute mode; None otherwise)."""
        return self._derived.w16.get(name) if self._derived is not None else None

    def _lstm_der(self, mname, l, bidirectional):
        d = self._derived.lstm
        return [d[f"{mname}.{l}.0"]] + ([d[f"{mname}.{l}.1"]] if bidirectional else [])

    # ---- frame-major building blocks
    def _check(self, x):
        if not x.is_cuda:
            raise RuntimeError("DisentangledVAE runs only on the MI355X HIP path; there is no CPU fallback "
                               "(the CPU restatement lives in oracle/ and is test infrastructure)")
        if x.shape[1] != N_MEL or x.shape[2] != self.n_frames:
            raise ValueError(f"expected [B, {N_MEL}, {self.n_frames}] mel segments, got {tuple(x.shape)}")

    def _lstm(self, mname, x, T, n_seg, x16=None):
        """x16: bf16 data of x when x is a placeholder (output of a conv block in the bf16 compute mode).
        Returns (h, h16) the same way: h16 is None unless the layer keeps its state in bf16 (then h is the placeholder)."""
        mod = getattr(self, mname)
        from ..derived import lstm_pack_modes
        # both passes on the W_hh-resident launches (bf16 mode): two plain layers; otherwise (fp32x3: forward only — inside
        # LstmStack2Fn — or none) two stacked layers of equal width share their per-frame launches
        mf, mb = lstm_pack_modes(ops.current_mode(), mod.hidden_size)
        ndir = 2 if mod.bidirectional else 1
        pers = ops.lstm_persistent_usable(n_seg, mod.hidden_size, mf, ndir) and \
            ops.lstm_persistent_usable(n_seg, mod.hidden_size, mb, ndir, bwd=True)
        if not pers and LstmStack2Fn.usable(T, mod.hidden_size, mod.num_layers, mod.bidirectional):
            # two stacked layers of equal width share their frame launches (ops.LstmStack2Fn)
            der = self._lstm_der(mname, 0, False) + self._lstm_der(mname, 1, False)
            return LstmStack2Fn.apply(x, T, n_seg, *mod.layer(0)[:4], *mod.layer(1)[:4], der, x16, True)
        for l in range(mod.num_layers):
            x, x16 = LstmLayerFn.apply(x, T, n_seg, *mod.layer(l), self._lstm_der(mname, l, mod.bidirectional), x16, True)
        return x, x16

    def _encode_frames(self, x, T, n_seg, groups):
        x16 = None
        for i, blk in enumerate(self.enc_modules):
            c, bn = _conv_of(blk), blk[1]
            x, x16 = ConvBnActFn.apply(x, c.weight, c.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var,
                                       bn.num_batches_tracked, None, n_seg, groups, ACT_RELU, self.training,
                                       self._wpt(f"enc_modules.{i}"), self._w16(f"enc_modules.{i}"), x16, True)
        h, _ = self._lstm("enc_lstm", x, T, n_seg, x16)                  # [T*N, 128]  (H = 64: fp32 state)
        d2 = 2 * self.dim_neck
        flat = Permute102Fn.apply(h, T, n_seg, d2, (n_seg, T * d2))     # index t*128+d as in :209
        lin = self.enc_linear.linear_layer
        feat = LinearFn.apply(flat, lin.weight, lin.bias, ACT_RELU, self._w16("enc_linear"))
        st, ct = self.style.linear_layer, self.content.linear_layer
        f_s, f_c = fanout(feat, 2)          # two consumers: their gradients are summed by one launch (ops.FanoutFn)
        return (LinearFn.apply(f_s, st.weight, st.bias, ACT_NONE, self._w16("style")),
                LinearFn.apply(f_c, ct.weight, ct.bias, ACT_NONE, self._w16("content")))

    def _decode_frames(self, z, T, n_seg, groups):
        p1, p2 = self.dec_pre_linear1, self.dec_pre_linear2
        h = LinearFn.apply(z, p1.weight, p1.bias, ACT_NONE, self._w16("dec_pre_linear1"))
        h = LinearFn.apply(h, p2.weight, p2.bias, ACT_NONE, self._w16("dec_pre_linear2"))   # [N, T*128]  (no activation, :232-233)
        d2 = 2 * self.dim_neck
        h = Permute102Fn.apply(h, n_seg, T, d2, (T * n_seg, d2))
        h, h16 = self._lstm("dec_lstm1", h, T, n_seg)
        for i, blk in enumerate(self.dec_modules):
            c, bn = blk[0], blk[1]
            h, h16 = ConvBnActFn.apply(h, c.weight, c.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var,
                                       bn.num_batches_tracked, None, n_seg, groups, ACT_RELU, self.training,
                                       self._wpt(f"dec_modules.{i}"), self._w16(f"dec_modules.{i}"), h16, True)
        h, h16 = self._lstm("dec_lstm2", h, T, n_seg, h16)
        lin = self.dec_linear2.linear_layer
        return LinearFn.apply(h, lin.weight, lin.bias, ACT_NONE, self._w16("dec_linear2"), h16)   # [T*N, 80]

    # ---- reference API
    def encode(self, x):
        """x [B,80,T] -> (style_mu, style_logvar, content_mu, content_logvar)  (disentangled_vae.py:198-220)"""
        self._check(x)
        self._refresh_derived()
        B, _, T = x.shape
        style, content = self._encode_frames(mel_to_frames(x.contiguous()), T, B, 1)
        s, c = self.speaker_size, self.latent_dim - self.speaker_size
        return style[:, :s], style[:, s:], content[:, :c], content[:, c:]

    def encode_heads(self, x):
        """x [B,80,T] -> (style [B, 2S] = mu|logvar, content [B, 2Cn] = mu|logvar): encode() without the column split."""
        self._check(x)
        self._refresh_derived()
        B, _, T = x.shape
        return self._encode_frames(mel_to_frames(x.contiguous()), T, B, 1)

    def _reparameterize(self, mu, logvar, train=True):
        """API-compatible standalone form (disentangled_vae.py:222-228).  The training path does not call it:
        forward() fuses the three reparameterisations into one HIP kernel (ops.LatentFn)."""
        if not train:
            return mu
        eps = torch.randn(logvar.shape, device=logvar.device, dtype=logvar.dtype)
        return eps * torch.exp(0.5 * logvar) + mu

    def decode(self, z):
        """z [B, latent] -> [B,80,T]  (disentangled_vae.py:230-248)"""
        if not z.is_cuda:
            raise RuntimeError("DisentangledVAE.decode runs only on the HIP device")
        self._refresh_derived()
        B, T = z.shape[0], self.n_frames
        y = self._decode_frames(z.contiguous(), T, B, 1)
        return FramesToMelFn.apply(y, B, N_MEL, T)

    def _eps(self, Bh, dev, train):
        S, Cn = self.speaker_size, self.latent_dim - self.speaker_size
        if self.eps_override is not None:
            e1, e2, es = (e.to(dev, torch.float32) for e in self.eps_override)
            if not train:
                eps_c = None
            elif (e1.is_contiguous() and e2.is_contiguous() and e1.shape == e2.shape and
                  e1.untyped_storage().data_ptr() == e2.untyped_storage().data_ptr() and
                  e2.storage_offset() == e1.storage_offset() + e1.numel()):
                eps_c = e1.as_strided((2 * e1.shape[0], e1.shape[1]), (e1.shape[1], 1))   # adjacent halves of one buffer
            else:
                eps_c = torch.cat((e1, e2), 0).contiguous()
            return eps_c, es.contiguous()
        eps_c = torch.randn((2 * Bh, Cn), device=dev, dtype=torch.float32) if train else None
        return eps_c, torch.randn((Bh, S), device=dev, dtype=torch.float32)

    def forward_full(self, x1, x2, train=True):
        """The forward pass with its outputs UNSPLIT: (rec, rec_hat, q_mu, q_lv, s_mu, s_lv) with rec / rec_hat [2*Bh, 80, T]
        and q_mu / q_lv [2*Bh, Cn] holding the x1 half first.  `forward` returns the reference's ten tensors as views of
        these; the training step hands the unsplit tensors to the fused loss (ops.LossGVAE2FullFn), so that autograd
        has no slice to undo (each `t[:Bh]` costs a zero-fill and a copy in backward)."""
        self._check(x1)
        self._check(x2)
        self._refresh_derived()
        Bh, _, T = x1.shape
        N = 2 * Bh
        S, Cn = self.speaker_size, self.latent_dim - self.speaker_size
        # [T*N, 80]; bf16 in the bf16 compute mode (it only feeds the first conv)
        x = mel_to_frames(x1.contiguous(), x2.contiguous(), dtype=ops.act_storage(ops.current_mode()))
        style, content = self._encode_frames(x, T, N, 2)
        eps_c, eps_s = self._eps(Bh, x1.device, train)
        z, q_mu, q_lv, s_mu, s_lv = LatentFn.apply(style, content, eps_c, eps_s, Bh, S, Cn)
        y = self._decode_frames(z, T, N, 2)                              # [T*N, 80]
        y, y_res, y_in = fanout(y, 3)       # loss, residual and postnet input: one gradient sum (ops.FanoutFn)
        y_hat = self.postnet.forward_frames(y_in, N, 2, residual=y_res,  # y + postnet(y)
                                            wpt=[self._wpt(f"postnet.{i}") for i in range(5)],
                                            w16=[self._w16(f"postnet.{i}") for i in range(5)])
        rec = FramesToMelFn.apply(y, N, N_MEL, T)
        rec_hat = FramesToMelFn.apply(y_hat, N, N_MEL, T)
        return rec, rec_hat, q_mu, q_lv, s_mu, s_lv

    def forward(self, x1, x2, train=True):
        """-> (recons_x1, recons_x2, recons_x1_hat, recons_x2_hat, q_z1_mu, q_z1_logvar, q_z2_mu, q_z2_logvar,
        z_style_mu, z_style_logvar)   (disentangled_vae.py:250-279)"""
        rec, rec_hat, q_mu, q_lv, s_mu, s_lv = self.forward_full(x1, x2, train)
        Bh = x1.shape[0]
        return (rec[:Bh], rec[Bh:], rec_hat[:Bh], rec_hat[Bh:], q_mu[:Bh], q_lv[:Bh], q_mu[Bh:], q_lv[Bh:],
                s_mu, s_lv)

    def reference_layout(self, name, tensor):
        """`tensor` shaped like parameter `name` (its gradient, a moment, ...) in the REFERENCE's layout: conv weights
        are stored packed [5][Cout][Cin] here and [Cout][Cin][5] in the reference; everything else is identical."""
        mod = self.get_submodule(name.rsplit(".", 1)[0])
        if isinstance(mod, _Conv1dParams) and name.endswith(".weight"):
            return mod.torch_layout(tensor)
        return tensor

    def store_first_params(self, batch_size):
        """The two [2048 x T*128] weights (enc_linear, dec_pre_linear2; 134 MB each at T = 128, 70 % of all parameters):
        their gradient is ONE outer product over the 2*batch rows of a step, written by an unsplit launch — stored, not
        accumulated (FlatAdam.set_store_first).  Only while that launch stays unsplit (a small contraction depth)."""
        from ..ops import _split_k, _tiles
        out = []
        for name, w in (("enc_linear.linear_layer.weight", self.enc_linear.linear_layer.weight),
                        ("dec_pre_linear2.weight", self.dec_pre_linear2.weight)):
            if _split_k(_tiles(w.shape[0], w.shape[1]), 2 * int(batch_size)) == 1:
                out.append(name)
        return out

    def storage_layout(self, name, tensor):
        """Inverse of reference_layout: a tensor in the reference's layout -> how parameter `name` is stored here."""
        mod = self.get_submodule(name.rsplit(".", 1)[0])
        if isinstance(mod, _Conv1dParams) and name.endswith(".weight"):
            return tensor.permute(2, 0, 1)
        return tensor

    def update_c(self):
        self._c += self._c_delta

    def update_beta(self):
        self._beta += self._beta_delta


class ConvolutionalMulVAE(VariationalBaseModelVAE):
    """reference: disentangled_vae.py:288-350 (model + Adam owner, loss_functionGVAE2)."""

    def __init__(self, dataset, width, height, latent_sz, learning_rate, alpha, log_interval, normalize, batch_size,
                 speaker_size, channels=1, device=torch.device("cuda"), latent_dim=256, beta=0.1, mse_cof=10,
                 kl_cof=10, style_cof=0.1, n_frames: Optional[int] = None):
        super().__init__(dataset, width, height, channels, latent_sz, learning_rate, device, log_interval, batch_size)
        self.batch_size = batch_size
        self.alpha = alpha
        self.lr = learning_rate
        self.latent_dim = latent_dim
        self.mse_cof, self.kl_cof, self.style_cof = mse_cof, kl_cof, style_cof
        n_frames = width if n_frames is None else n_frames
        self.model = DisentangledVAE(latent_dim=self.latent_dim, beta=0.1, batch_size=batch_size,
                                     speaker_size=speaker_size, n_frames=n_frames).to(device)
        self.optimizer = FlatAdam(self.model.backward_param_order(), lr=self.lr, layout=self.model)
        self.optimizer.set_store_first(self.model.store_first_params(self.batch_size))
        self.train_losses, self.test_losses = [], []

    def loss_functionGVAE2(self, x1, x2, x_recon1, x_recon2, recons_x1_hat, recons_x2_hat, q_z1_mu, q_z1_logvar,
                           q_z2_mu, q_z2_logvar, style_mu1, style_logvar1, train=False):
        """8-tuple (LOSS, L1_x1, L1_x2, L1_x1hat, L1_x2hat, KL_z1, KL_z2, z_kl_style), disentangled_vae.py:310-327.
        The variables are named MSE_* in the reference but are L1 sums divided by the CONFIGURED batch size."""
        return self.losses_vector(x1, x2, x_recon1, x_recon2, recons_x1_hat, recons_x2_hat, q_z1_mu, q_z1_logvar,
                                  q_z2_mu, q_z2_logvar, style_mu1, style_logvar1).unbind(0)

    def losses_vector_full(self, x1, x2, rec, rec_hat, q_mu, q_lv, s_mu, s_lv):
        """losses_vector on the unsplit outputs of `DisentangledVAE.forward_full` (the training step's path)."""
        inv_b = 1.0 / float(self.batch_size)
        return ops.LossGVAE2FullFn.apply(x1, x2, rec, rec_hat, q_mu, q_lv, s_mu, s_lv, inv_b, -0.5 / x1.shape[0], -inv_b,
                                         self.mse_cof, self.kl_cof)

    def losses_vector(self, x1, x2, x_recon1, x_recon2, recons_x1_hat, recons_x2_hat, q_z1_mu, q_z1_logvar, q_z2_mu,
                      q_z2_logvar, style_mu1, style_logvar1):
        """The same eight scalars as one device vector [8] (two launches forward, one backward: ops.LossGVAE2Fn)."""
        inv_b = 1.0 / float(self.batch_size)
        return ops.LossGVAE2Fn.apply(x1, x2, x_recon1, x_recon2, recons_x1_hat, recons_x2_hat, q_z1_mu, q_z1_logvar,
                                     q_z2_mu, q_z2_logvar, style_mu1, style_logvar1, inv_b, -0.5 / q_z1_mu.shape[0],
                                     -inv_b, self.mse_cof, self.kl_cof)

    def update_(self):
        self.model.update_c()
        self.model.update_beta()

    def update_kl(self):
        self.kl_cof = min(self.kl_cof * 2, 10)

    def set_kl(self, beta):
        self.kl = beta
