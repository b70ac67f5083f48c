"""Weight-derived operand layouts of a DisentangledVAE, refreshed by ONE HIP launch (csrc/repack.hip,
`dvae_repack_all`) at the start of every forward — i.e. once per training step, after Adam moved the weights:

  * Conv1d: the transposed pack Wpt[5][Cin][Cout] the data gradient reads k-contiguously (the weights themselves LIVE
    in the packed layout Wp[5][Cout][Cin], see model/disentangled_vae._Conv1dParams: no forward pack, and the weight
    gradient accumulates straight into the flat gradient buffer);
  * nn.LSTM: b_ih + b_hh, W_ih^T (input-projection data gradient), W_hh^T (backward recurrence at H = 64 / generic H)
    and, for H a multiple of 512, W_hh in MFMA fragment order for the forward and backward frame kernels (fp32, bf16 or
    three-plane bf16 fragments, as the compute mode in force says).

Round 1 produced these with one small launch per use (42 pack / transpose launches and 19 ATen bias adds per step,
each into a fresh torch.empty).  The buffers are persistent; the descriptor table is rebuilt only when a source
parameter has moved (FlatAdam re-homes parameters into its flat buffer after the model is built).
"""
from __future__ import annotations

from collections import namedtuple
from typing import Dict, List, Optional  # noqa: F401

import torch

from . import _lib
from ._lib import check, lib, stream

# w_ih_t: W_ih^T (fp32, or bf16 in the bf16 compute mode); w_ih16: bf16 copy of W_ih (bf16 mode only, else None)
LstmDerived = namedtuple("LstmDerived", "bias w_ih_t w_hh_t pack_f pack_b w_ih16 cat", defaults=(None, None))
# The two directions of a narrow BiLSTM layer (H = 64: the encoder) side by side, so that ONE contraction serves both:
# w_ih [8H, In] (fp32, or bf16 in the bf16 mode) for the input projections (N = 8H), bias [8H], w_ih_t [In, 8H] for the
# data gradient (K = 8H).  Shared by the two directions' LstmDerived entries (whose bias / w_ih16 are views into it).
BiCat = namedtuple("BiCat", "w_ih bias w_ih_t")


def lstm_pack_modes(mode: int, H: int):
    """(forward pack mode, backward pack mode) of an LSTM layer under compute mode `mode`: the bf16 mode runs both
    recurrences on bf16 fragments; fp32x3 runs both on three-plane fragments — fp32 results at 6/16 of the fp32-MFMA
    cycles: the W_hh-resident persistent kernels (csrc/lstm_pers.hip) are matrix-pipe bound, forward 7.1 us per H = 1024
    frame against 13.8 on the per-frame kernels, backward (dG split into planes by the consumer) 12.2 against 13.4 on
    resident fp32 fragments and 16.8-18.9 on the per-frame fp32 kernels.  H = 64 (whole-sequence kernels): the
    mode's own arithmetic (W_hh rounded / split in registers, no pack); other H not a multiple of 512: fp32."""
    if H == 64 and mode in (_lib.MODE_BF16, _lib.MODE_F32X3):
        return mode, mode
    if H % 512:
        return _lib.MODE_F32, _lib.MODE_F32
    if mode == _lib.MODE_BF16:
        return _lib.MODE_BF16, _lib.MODE_BF16
    if mode == _lib.MODE_F32X3:
        return _lib.MODE_F32X3, _lib.MODE_F32X3
    return _lib.MODE_F32, _lib.MODE_F32


def _desc(kind, src, dst, d0, d1=0, src2=None, dst2=None, d2=0):
    d = _lib.RepackDesc()
    d.kind, d.d0, d.d1, d.d2 = kind, d0, d1, d2
    d.src, d.src2 = src.data_ptr(), (src2.data_ptr() if src2 is not None else None)
    d.dst, d.dst2 = dst.data_ptr(), (dst2.data_ptr() if dst2 is not None else None)
    return d


def run_descs(descs: List):
    arr = (_lib.RepackDesc * len(descs))(*descs)
    check(lib().dvae_repack_all(arr, len(descs), stream()), "dvae_repack_all")


def conv_wpt_local(conv_wp: torch.Tensor) -> torch.Tensor:
    """Wp[5][Cout][Cin] -> Wpt[5][Cin][Cout] (stand-alone use of ConvBnActFn: kernel tests, Postnet on its own)."""
    _, cout, cin = conv_wp.shape
    wpt = torch.empty((5, cin, cout), device=conv_wp.device, dtype=torch.float32)
    run_descs([_desc(_lib.REPACK_CONV_T, conv_wp, wpt, cout, cin)])
    return wpt


def lstm_local(w_ih, w_hh, b_ih, b_hh, mode: int) -> LstmDerived:
    """The derived operands of one (layer, direction), computed on the spot (stand-alone use of LstmLayerFn)."""
    H, In = w_hh.shape[1], w_ih.shape[1]
    f = dict(device=w_hh.device, dtype=torch.float32)
    bias, wit, wht = torch.empty(4 * H, **f), torch.empty((In, 4 * H), **f), torch.empty((H, 4 * H), **f)
    descs = [_desc(_lib.REPACK_ADD2, b_ih, bias, 4 * H, src2=b_hh),
             _desc(_lib.REPACK_TRANSPOSE, w_ih, wit, 4 * H, In),
             _desc(_lib.REPACK_TRANSPOSE, w_hh, wht, 4 * H, H)]
    pf = pb = None
    if H % 512 == 0:
        pf, pb = torch.empty(6 * H * H, **f), torch.empty(6 * H * H, **f)      # 6 bytes per element (three bf16 planes)
        mf, mb = lstm_pack_modes(mode, H)
        descs.append(_desc(_lib.REPACK_LSTM_PACK, w_hh, pf, H, d1=mf, dst2=pb, d2=mb))
    run_descs(descs)
    return LstmDerived(bias, wit, wht, pf, pb)


def lstm_local_pair(params, mode: int):
    """Both directions of a narrow (H = 64) bidirectional layer, computed on the spot, with the side-by-side buffers
    (BiCat) the layer's merged contractions read.  params: [(w_ih, w_hh, b_ih, b_hh)] x 2."""
    H, In = params[0][1].shape[1], params[0][0].shape[1]
    dev = params[0][1].device
    f = dict(device=dev, dtype=torch.float32)
    b16 = mode == _lib.MODE_BF16
    wdt = torch.bfloat16 if b16 else torch.float32
    cat = BiCat(torch.empty((8 * H, In), device=dev, dtype=wdt), torch.empty(8 * H, **f),
                torch.empty((In, 8 * H), device=dev, dtype=wdt))
    out, descs = [], []
    for dn, (w_ih, w_hh, b_ih, b_hh) in enumerate(params):
        rows = slice(dn * 4 * H, (dn + 1) * 4 * H)
        d = LstmDerived(cat.bias[rows], None, torch.empty((H, 4 * H), **f), None, None, cat.w_ih[rows] if b16 else None, cat)
        descs += [_desc(_lib.REPACK_ADD2, b_ih, d.bias, 4 * H, src2=b_hh),
                  _desc(_lib.REPACK_TRANSPOSE, w_ih, cat.w_ih_t[:, rows], 4 * H, In, d2=int(b16) | ((8 * H) << 1)),
                  _desc(_lib.REPACK_CAST_BF16 if b16 else _lib.REPACK_COPY_F32, w_ih, cat.w_ih[rows], 4 * H * In, 1),
                  _desc(_lib.REPACK_TRANSPOSE, w_hh, d.w_hh_t, 4 * H, H)]
        out.append(d)
    run_descs(descs)
    return out


class DerivedWeights:
    def __init__(self, convs: Dict[str, torch.nn.Parameter], lstms: Dict[str, tuple],
                 casts: Optional[Dict[str, torch.nn.Parameter]] = None):
        """convs: name -> packed conv weight parameter that needs a data gradient; lstms: name -> (w_ih, w_hh, b_ih,
        b_hh) of one (layer, direction); casts: name -> weight (conv packs, Linear weights) whose bf16 copy is the B
        operand of a contraction in the bf16 compute mode."""
        self._convs, self._lstms, self._casts = convs, lstms, dict(casts or {})
        self.w16: Dict[str, torch.Tensor] = {}
        self._sig = None
        self._descs = None
        self.wpt: Dict[str, torch.Tensor] = {}
        self.lstm: Dict[str, LstmDerived] = {}
        self.refreshes = 0

    def _signature(self):
        return tuple(p.data_ptr() for p in self._convs.values()) + \
            tuple(p.data_ptr() for ps in self._lstms.values() for p in ps) + \
            tuple(p.data_ptr() for p in self._casts.values())

    def _build(self, mode):
        """(Re)build buffers and the descriptor table for compute mode `mode`.  In the bf16 mode the weight operands of
        the forward / data-gradient contractions are bf16 COPIES (half the bytes, no conversion in the kernels)."""
        descs = []
        b16 = mode == _lib.MODE_BF16
        wdt = torch.bfloat16 if b16 else torch.float32
        self.wpt, self.lstm, self.w16 = {}, {}, {}
        for name, wp in self._convs.items():
            _, cout, cin = wp.shape
            self.wpt[name] = torch.empty((5, cin, cout), device=wp.device, dtype=wdt)
            descs.append(_desc(_lib.REPACK_CONV_T, wp, self.wpt[name], cout, cin, d2=int(b16)))
        if b16:
            for name, w in self._casts.items():
                self.w16[name] = torch.empty(w.shape, device=w.device, dtype=torch.bfloat16)
                descs.append(_desc(_lib.REPACK_CAST_BF16, w, self.w16[name], w.numel(), 1))
        cats = {}
        for name, (w_ih, w_hh, b_ih, b_hh) in self._lstms.items():
            H, In = w_hh.shape[1], w_ih.shape[1]
            f = dict(device=w_hh.device, dtype=torch.float32)
            big = H % 512 == 0
            stem, dno = name.rsplit(".", 1)
            if H == 64 and f"{stem}.1" in self._lstms:
                # a direction of a narrow bidirectional layer: its bias / W_ih copy / W_ih^T are the halves of the pair's
                # side-by-side buffers
                if stem not in cats:
                    cats[stem] = BiCat(torch.empty((8 * H, In), device=w_hh.device, dtype=wdt), torch.empty(8 * H, **f),
                                       torch.empty((In, 8 * H), device=w_hh.device, dtype=wdt))
                cat, dn = cats[stem], int(dno)
                rows = slice(dn * 4 * H, (dn + 1) * 4 * H)
                d = LstmDerived(cat.bias[rows], None, torch.empty((H, 4 * H), **f), None, None,
                                cat.w_ih[rows] if b16 else None, cat)
                self.lstm[name] = d
                descs.append(_desc(_lib.REPACK_ADD2, b_ih, d.bias, 4 * H, src2=b_hh))
                descs.append(_desc(_lib.REPACK_TRANSPOSE, w_ih, cat.w_ih_t[:, rows], 4 * H, In, d2=int(b16) | ((8 * H) << 1)))
                descs.append(_desc(_lib.REPACK_CAST_BF16 if b16 else _lib.REPACK_COPY_F32, w_ih, cat.w_ih[rows],
                                   4 * H * In, 1))
                descs.append(_desc(_lib.REPACK_TRANSPOSE, w_hh, d.w_hh_t, 4 * H, H))
                continue
            d = LstmDerived(torch.empty(4 * H, **f), torch.empty((In, 4 * H), device=w_hh.device, dtype=wdt),
                            torch.empty((H, 4 * H), **f),
                            torch.empty(6 * H * H, **f) if big else None,     # up to 3 bf16 planes
                            torch.empty(6 * H * H, **f) if big else None,
                            torch.empty((4 * H, In), device=w_hh.device, dtype=torch.bfloat16) if b16 else None)
            self.lstm[name] = d
            descs.append(_desc(_lib.REPACK_ADD2, b_ih, d.bias, 4 * H, src2=b_hh))
            descs.append(_desc(_lib.REPACK_TRANSPOSE, w_ih, d.w_ih_t, 4 * H, In, d2=int(b16)))
            if b16:
                descs.append(_desc(_lib.REPACK_CAST_BF16, w_ih, d.w_ih16, 4 * H * In, 1))
            descs.append(_desc(_lib.REPACK_TRANSPOSE, w_hh, d.w_hh_t, 4 * H, H))
            if d.pack_f is not None:
                mf, mb = lstm_pack_modes(mode, H)
                descs.append(_desc(_lib.REPACK_LSTM_PACK, w_hh, d.pack_f, H, d1=mf, dst2=d.pack_b, d2=mb))
        self._descs = (_lib.RepackDesc * len(descs))(*descs)

    def refresh(self, mode: int):
        """One launch: every derived buffer from the current weights (asynchronous on the current stream)."""
        sig = self._signature() + (int(mode),)
        if sig != self._sig:
            self._build(int(mode))
            self._sig = sig
        check(lib().dvae_repack_all(self._descs, len(self._descs), stream()), "dvae_repack_all")
        self.refreshes += 1
