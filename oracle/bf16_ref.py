"""TEST INFRASTRUCTURE ONLY — CPU oracle of the **bf16 compute mode** (BASELINE configs[2] / [4]).

The reference itself has no bf16 path (it is plain fp32 `torch.nn`), so there is nothing of the
reference to pin this file against: it is the fp32 oracle (oracle/dvae_ref.py, pinned to the reference's golden
vectors) with ONE change, stated operation by operation, that defines what "bf16 compute" means here:

    every contraction rounds BOTH operands to bf16 (round-to-nearest-even) and accumulates in fp32 —
    forward products, data gradients and weight gradients alike; nothing else is rounded.

      Linear / LSTM input projection   y  = r(x) r(W)^T + b        dx = r(dy) r(W)     dW = r(dy)^T r(x)
      Conv1d (k5)                      y  = conv(r(x), r(w)) + b   dx = conv^T(r(dy), r(w))   dw = corr(r(dy), r(x))
      LSTM recurrence (every H)        g_t = pre_t + r(h_{t-1}) r(W_hh)^T      dh_{t-1} = r(dg_t) r(W_hh)
      dW_hh (every H)                  r(dg)^T r(h)

Tensors between operations (activations, gates, BatchNorm, losses, gradients, master weights, Adam) are fp32, with one
exception that is visible in the results: the wide LSTM layers (H % 512 == 0) keep their OUTPUT h_t and their gate
gradient dg_t in bf16 (the frame kernels write what the next frame's contraction reads).  Every consumer of h_t is a
contraction, which would round it anyway; dg_t also feeds the bias gradient, which therefore sums r(dg_t):

      LSTM, H % 512 == 0               h_t := r(h_t) as stored       db_ih = db_hh = sum_t r(dg_t)
A product of two bf16 numbers is exact in fp32, so the HIP path and this oracle differ only by fp32 summation order
(and by the occasional operand that such a difference pushes across a bf16 rounding boundary).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from .dvae_ref import RefDVAE


def r16(t: torch.Tensor) -> torch.Tensor:
    return t.bfloat16().to(t.dtype)


class _MatmulNT(torch.autograd.Function):
    """y[..., N] = op(x)[..., K] @ op(w)[N, K]^T;  op = r16 when `round_fwd`, identity otherwise; the weight gradient
    always uses rounded operands (it is a GEMM launch in the HIP path whatever H is)."""

    @staticmethod
    def forward(ctx, x, w, round_fwd=True):
        xr, wr = (r16(x), r16(w)) if round_fwd else (x, w)
        ctx.save_for_backward(x, w)
        ctx.round_fwd = round_fwd
        return xr @ wr.t()

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dyr = r16(dy)
        dx = (dyr @ r16(w)) if ctx.round_fwd else (dy @ w)
        dw = dyr.reshape(-1, dyr.shape[-1]).t() @ r16(x).reshape(-1, x.shape[-1])
        return dx, dw, None


class _Conv5(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w):
        ctx.save_for_backward(x, w)
        return F.conv1d(r16(x), r16(w), None, padding=2)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dyr = r16(dy)
        dx = torch.nn.grad.conv1d_input(x.shape, r16(w), dyr, padding=2)
        dw = torch.nn.grad.conv1d_weight(r16(x), w.shape, dyr, padding=2)
        return dx, dw


def linear(x, weight, bias):
    return _MatmulNT.apply(x, weight, True) + bias


def conv5(x, weight, bias):
    return _Conv5.apply(x, weight) + bias[None, :, None]


class _RoundFwd(torch.autograd.Function):
    """r16 forward, identity backward (a tensor STORED in bf16 whose gradient travels in fp32)"""

    @staticmethod
    def forward(ctx, t):
        return r16(t)

    @staticmethod
    def backward(ctx, g):
        return g


class _RoundBwd(torch.autograd.Function):
    """identity forward, r16 backward (a GRADIENT stored in bf16)"""

    @staticmethod
    def forward(ctx, t):
        return t.view_as(t)

    @staticmethod
    def backward(ctx, g):
        return r16(g)


def lstm_dir(x, w_ih, w_hh, b_ih, b_hh, reverse=False, state_bf16=None):
    """x [B, T, In] -> h [B, T, H]; gate order i, f, g, o; zero initial state (nn.LSTM semantics).
    state_bf16 (default: H % 512 == 0): h_t and dg_t are stored in bf16, see the header."""
    B, T, _ = x.shape
    H = w_hh.shape[1]
    state_bf16 = (H % 512 == 0) if state_bf16 is None else state_bf16
    pre = _MatmulNT.apply(x, w_ih, True) + (b_ih + b_hh)
    h = x.new_zeros(B, H)
    c = x.new_zeros(B, H)
    outs = [None] * T
    order = range(T - 1, -1, -1) if reverse else range(T)
    first = True
    for t in order:
        g = pre[:, t]
        if not first:
            g = g + _MatmulNT.apply(h, w_hh, True)
        first = False
        if state_bf16:
            g = _RoundBwd.apply(g)
        i, f, gg, o = g.chunk(4, dim=1)
        c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
        h = torch.sigmoid(o) * torch.tanh(c)
        if state_bf16:
            h = _RoundFwd.apply(h)
        outs[t] = h
    return torch.stack(outs, dim=1)


def lstm(mod: torch.nn.LSTM, x):
    for layer in range(mod.num_layers):
        p = lambda n, sfx="": getattr(mod, f"{n}_l{layer}{sfx}")
        out = lstm_dir(x, p("weight_ih"), p("weight_hh"), p("bias_ih"), p("bias_hh"))
        if mod.bidirectional:
            rev = lstm_dir(x, p("weight_ih", "_reverse"), p("weight_hh", "_reverse"), p("bias_ih", "_reverse"),
                           p("bias_hh", "_reverse"), reverse=True)
            out = torch.cat((out, rev), dim=-1)
        x = out
    return x


def _child(m):
    return getattr(m, m._attr) if hasattr(m, "_attr") else m


def _conv_bn(blk, x):
    conv, bn = _child(blk[0]), blk[1]
    return bn(conv5(x, conv.weight, conv.bias))


def _lin(m, x):
    m = _child(m)
    return linear(x, m.weight, m.bias)


class RefDVAEBf16(RefDVAE):
    """Same module tree / state_dict keys / forward() as RefDVAE; only the contractions differ (see the header)."""

    def encode(self, x):
        nb = x.shape[0]
        for blk in self.enc_modules:
            x = F.relu(_conv_bn(blk, x))
        seq = lstm(self.enc_lstm, x.transpose(1, 2))
        feat = F.relu(_lin(self.enc_linear, seq.reshape(nb, -1)))
        st, ct = _lin(self.style, feat), _lin(self.content, feat)
        s, c = self.speaker_size, self.latent_dim - self.speaker_size
        return st[:, :s], st[:, s:], ct[:, :c], ct[:, c:]

    def decode(self, z):
        h = _lin(self.dec_pre_linear2, _lin(self.dec_pre_linear1, z))
        h = h.view(z.shape[0], -1, 2 * self.dim_neck)
        h = lstm(self.dec_lstm1, h).transpose(1, 2)
        for blk in self.dec_modules:
            h = F.relu(_conv_bn(blk, h))
        h = lstm(self.dec_lstm2, h.transpose(1, 2))
        return _lin(self.dec_linear2, h).transpose(1, 2)

    def postnet_fwd(self, x):
        convs = self.postnet.convolutions
        for blk in convs[:-1]:
            x = torch.tanh(_conv_bn(blk, x))
        return _conv_bn(convs[-1], x)

    def forward(self, x1, x2, eps):
        post = self.postnet.forward
        self.postnet.forward = self.postnet_fwd       # RefDVAE.forward calls self.postnet(...)
        try:
            return super().forward(x1, x2, eps)
        finally:
            self.postnet.forward = post
