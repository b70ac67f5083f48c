"""Flat-buffer Adam: all parameters live in ONE contiguous fp32 buffer (each parameter a 16-byte
aligned view), likewise gradients and both moments, so the optimiser is a single HBM-bound HIP
launch (dvae_adam_flat_dev: 7 x 4 bytes per parameter) and a data-parallel all-reduce runs over
contiguous slices of the gradient buffer without any packing copy.

Everything that changes between steps lives ON THE DEVICE (`dev_state`: step count, bias corrections, learning rate,
gradient scale), so a captured hipGraph replays a correct step and `param_groups[0]["lr"] = ...` (a schedule) needs no
re-capture: `sync_scalars()` — called by `step()` outside a capture and by the trainer before every replay — copies the
host values over when they changed.  The same launch clears the gradient ranges `zero_grad()` covers after reading them
(the next step's zero_grad is then free), and it does NOTHING while the sticky error word of the persistent LSTM launches
is non-zero: a recurrence that gave up a bounded wait leaves garbage gradients, and weights / moments must not consume
them before the host has looked (`ops.lstm_pers_check`).

Replaces torch.optim.Adam(self.model.parameters(), lr) at /root/reference/model/disentangled_vae.py:304
(betas (0.9, 0.999), eps 1e-8, no weight decay, no amsgrad).
"""
from __future__ import annotations

from typing import Dict, Iterable, List

import torch

from ._lib import check, lib, ptr, stream

_ALIGN = 4  # elements (16 bytes)
_PAD = 32   # the flat buffers' length is a multiple of this: any world size up to 8 cuts it into 16-byte aligned shards


class FlatAdam:
    def __init__(self, named_params: Iterable, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, layout=None):
        """layout: object with reference_layout(name, t) / storage_layout(name, t) (the model): how a tensor shaped like
        parameter `name` maps between its STORAGE here and the reference's layout (conv weights are stored packed).
        Checkpointed moments are kept in the reference's layout, so they do not depend on how parameters are stored."""
        self.layout = layout
        items = list(named_params)
        if items and not isinstance(items[0], (tuple, list)):
            items = [(f"p{i}", p) for i, p in enumerate(items)]
        self.names: List[str] = [n for n, _ in items]
        self.params: List[torch.nn.Parameter] = [p for _, p in items]
        if not self.params:
            raise ValueError("FlatAdam got an empty parameter list")
        dev = self.params[0].device
        self.offsets: Dict[str, int] = {}
        off = 0
        for n, p in items:
            if p.dtype != torch.float32 or p.device != dev:
                raise ValueError("FlatAdam needs fp32 parameters on one device")
            self.offsets[n] = off
            off += (p.numel() + _ALIGN - 1) // _ALIGN * _ALIGN
        self.n_used = off                                  # elements that belong to parameters
        off = (off + _PAD - 1) // _PAD * _PAD              # zero tail: a sharded step cuts the buffers into equal parts
        self.numel = off
        self.flat_p = torch.zeros(off, device=dev, dtype=torch.float32)
        self.flat_g = torch.zeros(off, device=dev, dtype=torch.float32)
        self.exp_avg = torch.zeros(off, device=dev, dtype=torch.float32)
        self.exp_avg_sq = torch.zeros(off, device=dev, dtype=torch.float32)
        for n, p in items:
            o = self.offsets[n]
            view = self.flat_p[o:o + p.numel()].view_as(p)
            view.copy_(p.data)
            p.data = view
            p.grad = self.flat_g[o:o + p.numel()].view_as(p)
            p._dvae_flat_owned = True      # ops._grad_buf refuses to re-allocate a gradient for such a parameter
            p._dvae_owner = self           # ... and marks this optimizer's gradient buffer as written to
        self.lr, self.betas, self.eps = lr, betas, eps
        self._zero_ranges = [(0, off)]     # what zero_grad clears: everything, minus store-first parameters
        # [t, 1-b1^t, sqrt(1-b2^t), -, lr, grad_scale, -, -]
        self.dev_state = torch.zeros(8, device=dev, dtype=torch.float32)
        self._dev_scalars = None           # (lr, grad_scale) as last written to dev_state[4:6]
        self._pin, self._pin_ev, self._pin_i = None, None, 0      # pinned staging slots of sync_scalars
        # True: the zero_grad ranges of flat_g are known to be zero (the last Adam launch cleared them after reading and no
        # backward kernel has accumulated since: ops._grad_buf resets it) — zero_grad() is then free
        self._clean = False
        self.fold_zero_grad = True         # let the Adam launch clear them (False: zero_grad launches every time)
        self.guard_device_errors = True    # skip the update while a persistent LSTM launch's error word is set
        # torch.optim-compatible surface used by callers of the reference wrapper
        self.param_groups = [{"params": self.params, "lr": lr, "betas": betas, "eps": eps}]

    def views_intact(self) -> bool:
        base = self.flat_p.data_ptr()
        for n, p in zip(self.names, self.params):
            if p.data_ptr() != base + 4 * self.offsets[n]:
                return False
            if p.grad is None or p.grad.data_ptr() != self.flat_g.data_ptr() + 4 * self.offsets[n]:
                return False
        return True

    def set_store_first(self, names):
        """Parameters whose gradient is WRITTEN (not accumulated) by exactly one launch per step
        (ops.linear_wgrad_acc(store=True)): zero_grad leaves their slices alone — the zero-fill and the read half of the
        read-modify-write of a 134 MB gradient are pure HBM traffic.  Only for parameters that get a gradient EVERY step."""
        names = set(names)
        for n, p in zip(self.names, self.params):
            p._dvae_grad_store_first = n in names
        spans = sorted((self.offsets[n], self.offsets[n] + (p.numel() + _ALIGN - 1) // _ALIGN * _ALIGN)
                       for n, p in zip(self.names, self.params) if n in names)
        out, lo = [], 0
        for a, b in spans:
            if a > lo:
                out.append((lo, a))
            lo = max(lo, b)
        if lo < self.numel:
            out.append((lo, self.numel))
        self._zero_ranges = out
        self._clean = False

    def zero_grad(self, set_to_none: bool = False):
        # gradients are accumulated by the HIP backward kernels directly into flat_g
        # a step starts here: the written-exactly-once count of the store-first gradients starts from zero, whatever an
        # abandoned attempt (a capture that failed after backward, an exception between backward and step) left behind
        for p in self.params:
            if getattr(p, "_dvae_grad_store_first", False):
                p._dvae_sf_writes = 0
        self.__dict__.get("_slab_pending", {}).clear()      # slabs of an abandoned backward pass are not this step's
        if self._clean:
            return      # the previous step's Adam launch cleared these ranges and nothing has accumulated since
        for lo, hi in self._zero_ranges:
            if self.flat_g.is_cuda:
                check(lib().dvae_zero_f32(self.flat_g.data_ptr() + 4 * lo, hi - lo, stream()), "dvae_zero_f32")
            else:
                self.flat_g[lo:hi].zero_()
        self._clean = self.flat_g.is_cuda and not torch.cuda.is_current_stream_capturing()

    def sync_scalars(self, grad_scale: float = None):
        """Host -> device copy of (lr, grad_scale) when they changed since the last copy.  NOT capturable by design: the
        trainer calls it before capture / replay, `step()` calls it itself when the stream is not capturing."""
        lr = float(self.param_groups[0]["lr"])
        gs = float(self._dev_scalars[1] if (grad_scale is None and self._dev_scalars) else (grad_scale or 1.0))
        if self._dev_scalars != (lr, gs):
            if self.dev_state.is_cuda:
                # staged through a small ring of PINNED slots, asynchronously: a per-step schedule must not stall the host
                # behind a pageable copy every step; a slot is reused only after the copy that read it has completed
                if self._pin is None:
                    self._pin = torch.empty(16, 2, dtype=torch.float32, pin_memory=True)
                    self._pin_ev = [None] * 16
                i = self._pin_i = (self._pin_i + 1) % 16
                if self._pin_ev[i] is not None:
                    self._pin_ev[i].synchronize()
                self._pin[i, 0], self._pin[i, 1] = lr, gs
                self.dev_state[4:6].copy_(self._pin[i], non_blocking=True)
                ev = torch.cuda.Event()
                ev.record()
                self._pin_ev[i] = ev
            else:
                self.dev_state[4:6].copy_(torch.tensor([lr, gs], dtype=torch.float32))
            self._dev_scalars = (lr, gs)

    def _store_first_guard(self):
        """A store-first parameter is excluded from zero_grad: its gradient must have been WRITTEN exactly once since the
        last step (ops.LinearFn.backward counts), else Adam would consume a stale or a partial gradient."""
        bad = None
        for n, p in zip(self.names, self.params):
            if getattr(p, "_dvae_grad_store_first", False):
                w = getattr(p, "_dvae_sf_writes", None)
                if w is not None and w != 1 and bad is None:
                    bad = (n, w)
                p._dvae_sf_writes = 0
        if bad is not None:
            raise RuntimeError(f"FlatAdam: store-first gradient of {bad[0]} was written {bad[1]} times since the last step "
                               "(exactly one backward pass per step; for gradient accumulation clear the flag with "
                               "set_store_first(()) first)")

    def step(self, grad_scale: float = 1.0):
        """One Adam step over all parameters: one launch."""
        self.step_range(0, self.numel, grad_scale, tick=True)

    def step_range(self, lo: int, hi: int, grad_scale: float = 1.0, tick: bool = True):
        """The update of elements [lo, hi) of the flat buffers.  `tick` advances the step counter first: exactly one call
        per step has it set (the first).  A sharded optimizer (ddp.GradReducer(mode="rs_ag")) calls this once per bucket
        on the slice this rank owns."""
        if not self.flat_p.is_cuda:
            raise RuntimeError("FlatAdam.step runs only on the HIP device (no CPU fallback)")
        from . import ops
        from ._lib import Ranges
        import ctypes as C
        if lo % 4 or hi % 4 or not 0 <= lo < hi <= self.numel:
            raise ValueError(f"FlatAdam.step_range: [{lo}, {hi}) is not a 16-byte aligned range of the flat buffers")
        if tick:
            if not self.views_intact():
                # model.zero_grad() (set_to_none), .to()/.float() or `p.grad = None` would detach parameters from the flat
                # buffers: the kernels would then accumulate elsewhere while Adam and the all-reduce read stale zeros
                raise RuntimeError("FlatAdam: a parameter or its .grad is no longer a view of the flat buffers "
                                   "(use optimizer.zero_grad(), never model.zero_grad()/p.grad = None/model.to())")
            self._store_first_guard()
            ops.join_side()     # weight-gradient work may still be running on the side stream
            ops.fold_pending(self)      # the k-split slabs of this step's weight gradients: one table-driven launch, fixed order
            if torch.cuda.is_current_stream_capturing():
                if self._dev_scalars is None or self._dev_scalars[1] != float(grad_scale):
                    raise RuntimeError("FlatAdam.step under capture: call sync_scalars(grad_scale) before the capture")
            else:
                self.sync_scalars(grad_scale)
        skip = None
        if self.guard_device_errors and ops.LSTM_PERSISTENT:
            skip = lib().dvae_lstm_pers_err_word(ptr(ops.lstm_pers_workspace(self.flat_p.device)))
        rg = Ranges()
        if self.fold_zero_grad:
            spans = [(max(a, lo) - lo, min(b, hi) - lo) for a, b in self._zero_ranges if min(b, hi) > max(a, lo)]
            if len(spans) > 8:
                raise RuntimeError("FlatAdam: more than 8 zero_grad ranges")
            rg.n = len(spans)
            for i, (a, b) in enumerate(spans):
                rg.lo[i], rg.hi[i] = a, b
        o = 4 * lo
        check(lib().dvae_adam_flat_dev(self.flat_p.data_ptr() + o, self.flat_g.data_ptr() + o, self.exp_avg.data_ptr() + o,
                                       self.exp_avg_sq.data_ptr() + o, hi - lo, self.betas[0], self.betas[1], self.eps,
                                       ptr(self.dev_state), skip, C.byref(rg), int(tick), stream()), "dvae_adam_flat_dev")
        # the whole buffer was read and cleared only by a full step; a sharded step leaves the other ranks' slices
        # untouched, and says so itself (GradReducer.step)
        self._clean = bool(self.fold_zero_grad) and lo == 0 and hi == self.numel

    @property
    def t(self) -> int:
        """Number of optimiser steps taken (kept on the device so a captured graph can advance it)."""
        return int(self.dev_state[0].item())

    STATE_FORMAT = 2      # 2: moments per parameter, in the REFERENCE's tensor layout (1 / absent: raw flat vectors)

    def _moment_views(self, flat):
        for n, p in zip(self.names, self.params):
            o = self.offsets[n]
            yield n, flat[o:o + p.numel()].view_as(p)

    def _to_ref(self, name, t):
        return self.layout.reference_layout(name, t) if self.layout is not None else t

    def _from_ref(self, name, t):
        return self.layout.storage_layout(name, t) if self.layout is not None else t

    def state_dict(self):
        """Moments per parameter in the reference's layout (what torch.optim.Adam would hold for the reference model):
        independent of the packed conv-weight storage and of the order of the flat buffer."""
        return {"format": self.STATE_FORMAT, "t": self.t, "lr": self.param_groups[0]["lr"], "betas": self.betas,
                "eps": self.eps, "names": list(self.names),
                "exp_avg": {n: self._to_ref(n, v).detach().cpu().contiguous() for n, v in self._moment_views(self.exp_avg)},
                "exp_avg_sq": {n: self._to_ref(n, v).detach().cpu().contiguous()
                               for n, v in self._moment_views(self.exp_avg_sq)}}

    def load_state_dict(self, sd, legacy_layout=None):
        """legacy_layout: only for format-1 files (see below): "packed" | "torch"."""
        if sorted(sd["names"]) != sorted(self.names):
            raise ValueError("optimizer state was saved for a different set of parameters")
        fmt = int(sd.get("format", 1))
        if fmt == 1:
            # raw flat vectors in the order of ITS `names`.  Two builds wrote this format and nothing in the file tells
            # them apart: before conv weights were stored packed every slice was in torch's layout [Cout][Cin][5]; the
            # build right before format 2 already stored (and saved) conv slices packed [5][Cout][Cin].  Guessing would
            # silently scramble every conv moment, so the caller has to say which one it is.
            if list(sd["names"]) != self.names:
                raise ValueError("format-1 optimizer state needs the parameter order it was saved with")
            if legacy_layout not in ("packed", "torch"):
                differs = [n for n, p in zip(self.names, self.params) if tuple(self._to_ref(n, p).shape) != tuple(p.shape)]
                if differs:
                    raise ValueError("format-1 optimizer state does not record how conv-weight slices are laid out: pass "
                                     "legacy_layout='packed' (files written by the build that stored conv weights "
                                     "[5][Cout][Cin]) or 'torch' ([Cout][Cin][5]); env DVAE_OPT_LEGACY_LAYOUT for "
                                     f"load_last_model.  Affected: {differs[:3]} ...")
            per = {}
            for key in ("exp_avg", "exp_avg_sq"):
                flat, d = sd[key], {}
                for n, p in zip(self.names, self.params):
                    o = self.offsets[n]
                    sl = flat[o:o + p.numel()]
                    if legacy_layout == "packed":
                        d[n] = self._to_ref(n, sl.view_as(p))          # slice in STORAGE layout -> reference layout
                    else:
                        d[n] = sl.view(tuple(self._to_ref(n, p).shape))
                per[key] = d
        elif fmt == self.STATE_FORMAT:
            per = {"exp_avg": sd["exp_avg"], "exp_avg_sq": sd["exp_avg_sq"]}
        else:
            raise ValueError(f"optimizer state format {fmt} is newer than this build understands ({self.STATE_FORMAT})")
        self.dev_state.zero_()
        self.dev_state[0] = float(sd["t"])
        self._dev_scalars = None           # lr / grad_scale are re-sent by the next sync_scalars
        self.param_groups[0]["lr"] = float(sd["lr"])
        for key, flat in (("exp_avg", self.exp_avg), ("exp_avg_sq", self.exp_avg_sq)):
            for n, v in self._moment_views(flat):
                src = per[key][n]
                want = tuple(self._to_ref(n, v).shape)
                if tuple(src.shape) != want:
                    raise ValueError(f"optimizer state of {n}: shape {tuple(src.shape)}, expected {want}")
                v.copy_(self._from_ref(n, src.to(v.device)))
