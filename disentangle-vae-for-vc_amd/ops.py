"""Autograd bindings of the HIP kernels (forward AND backward are hand-written HIP; autograd only
routes tensors between them).  Every tensor is fp32, contiguous, on the GPU; activations are
frame-major [T*N, C] (see include/dvae_hip.h).

Weight gradients are ACCUMULATED by the kernels straight into `param.grad` (views of one flat
buffer owned by the optimiser, see optim.py) — the backward functions return None for parameters.
`grad_ready_hook`, when set, is called with the parameter as soon as its gradient is complete
(used by ddp.py to overlap the bucketed all-reduce with the rest of backward).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Callable, Optional

import torch

from . import _lib
from ._lib import check, lib, ptr, stream

ACT_NONE, ACT_RELU, ACT_TANH = 0, 1, 2
EPI_STORE, EPI_ACCUM, EPI_ATOMIC = 0, 1, 2
BN_EPS, BN_MOMENTUM = 1e-5, 0.1

grad_ready_hook: Optional[Callable[[torch.Tensor], None]] = None

# Weight-gradient contractions do not feed the backward chain, so they run on a second HIP stream and fill the
# matrix pipes that the latency-bound LSTM frame kernels of the NEXT layer's backward leave idle.
# FlatAdam.step / GradReducer join the stream before they read gradients.  Opt-in with DVAE_SIDE_STREAM=1.
USE_SIDE_STREAM = os.environ.get("DVAE_SIDE_STREAM", "0") != "0"   # measured gain 1.5 %: off by default
_side_stream = None
_side_dirty = False
_cb_queued = False


_fold_owners: list = []      # optimisers with k-split slabs pending from the running backward pass (see _defer_fold)


def _end_of_backward():
    # autograd final callback: runs on the thread that called backward(), after the last node
    global _cb_queued
    _cb_queued = False
    join_side()
    # the k-split slabs of this pass's weight gradients: ONE table-driven launch, so that `.grad` is complete when backward()
    # returns (a reducer that issued from its hooks has summed its own already)
    while _fold_owners:
        fold_pending(_fold_owners.pop())


def _queue_end_of_backward() -> bool:
    """True when the final callback is (now) queued on the running backward pass; False outside one"""
    global _cb_queued
    if _cb_queued:
        return True
    try:
        torch.autograd.Variable._execution_engine.queue_callback(_end_of_backward)
        _cb_queued = True
    except RuntimeError:
        return False
    return True


def side_stream():
    global _side_stream
    if _side_stream is None:
        _side_stream = torch.cuda.Stream()
    return _side_stream


class side_work:
    """`with side_work(t1, t2, ...):` runs the enclosed launches on the side stream, ordered after everything
    already enqueued on the current stream; the listed tensors are kept alive for the side stream."""

    def __init__(self, *tensors):
        self.tensors = [t for t in tensors if t is not None]
        self.ctx = None

    def __enter__(self):
        global _side_dirty, _cb_queued
        if not USE_SIDE_STREAM:
            return self
        if not _cb_queued:
            # gradients must be visible to the caller's stream when backward() returns
            try:
                torch.autograd.Variable._execution_engine.queue_callback(_end_of_backward)
                _cb_queued = True
            except RuntimeError:
                pass   # not inside a backward pass: the caller joins explicitly (join_side)
        side = side_stream()
        side.wait_stream(torch.cuda.current_stream())
        self.ctx = torch.cuda.stream(side)
        self.ctx.__enter__()
        _side_dirty = True
        return self

    def __exit__(self, *exc):
        if self.ctx is not None:
            self.ctx.__exit__(*exc)
            for t in self.tensors:
                t.record_stream(_side_stream)
        return False


def join_side():
    """Make the current stream wait for all side-stream gradient work enqueued so far."""
    global _side_dirty
    if _side_dirty and _side_stream is not None:
        torch.cuda.current_stream().wait_stream(_side_stream)
        _side_dirty = False


def _ok(*ts, act16=()):
    """Every tensor fp32, contiguous, on the GPU; the tensors listed in `act16` (activations that are contraction
    operands) may also be bf16 — the storage their producers use in the bf16 compute mode."""
    for t in ts:
        if t is None:
            continue
        dts = (torch.float32, torch.bfloat16) if any(t is a for a in act16) else (torch.float32,)
        if not (t.is_cuda and t.dtype in dts and t.is_contiguous()):
            raise ValueError(f"HIP op needs contiguous fp32 GPU tensors, got {t.device} {t.dtype} "
                             f"contiguous={t.is_contiguous()}")


def _b16(t) -> int:
    return int(t is not None and t.dtype == torch.bfloat16)


def act_storage(mode) -> torch.dtype:
    """Storage of the activations that only feed contractions (conv-block outputs, BatchNorm data gradients): bf16 in
    the bf16 compute mode — written so by their producers — fp32 otherwise."""
    return torch.bfloat16 if int(mode) == MODE_BF16 else torch.float32


def _grad_buf(p: torch.Tensor) -> torch.Tensor:
    owner = getattr(p, "_dvae_owner", None)
    if owner is not None:
        owner._clean = False              # a backward kernel is about to write into the flat gradient buffer
    if p.grad is None:
        if getattr(p, "_dvae_flat_owned", False):
            raise RuntimeError("gradient of a FlatAdam-owned parameter was set to None (model.zero_grad()?): its "
                               "gradient must stay a view of the optimiser's flat buffer; use optimizer.zero_grad()")
        p.grad = torch.zeros_like(p)      # stand-alone use of an op (kernel tests): a private gradient buffer
    return p.grad


fold_on_ready = True      # data parallel: sum a gradient's slabs when it is reported ready (a reducer that issues its
                          # collectives only after backward clears this and sums everything once: GradReducer.finish)


def _ready(*ps):
    if grad_ready_hook is not None:
        join_side()
        if fold_on_ready:
            fold_pending(params=[p for p in ps if p is not None])      # the collective reads the SUMMED gradient
        for p in ps:
            if p is not None:
                grad_ready_hook(p)


# ----------------------------------------------------------------------------- k-split slabs (no atomics)
# A contraction that is cut along k stores every split's partial product plainly — split 0 into the result, the others into
# slabs — and the slabs are added in a fixed order: every element has one writer and one summation order, so a train step is
# run-to-run bit-identical in the DEFAULT mode (round 6; until then only under set_deterministic, which ran unsplit and
# slower).  Weight gradients of an optimiser-owned parameter are summed LATER, by one table-driven launch at the end of the
# backward pass (autograd's final callback; FlatAdam.step sums whatever is still pending) or — data parallel with
# collectives issued from the hooks — when their bucket becomes ready (_ready); everything else (Linear outputs and
# data gradients with few rows, stand-alone use in kernel tests) right behind the contraction (dvae_slab_sum, which also
# applies the activation).
_slab_param: dict = {}       # (data_ptr of the result, elements, cap) -> slab tensor of a gradient WITHOUT an owning optimiser; the
                             # slabs of an optimiser-owned gradient live in that optimiser (`_slab_store`) and die with it
_slab_scratch: dict = {}     # (device index, stream handle) -> scratch slabs of transient results (consumed on that stream)
_colsum_ws: dict = {}        # (device index, stream handle) -> workspace of the deterministic column sums
SLAB_CAP = 16                # slabs provided to a split contraction = the most k-splits it may take (the 256-row tiles cut K
                             # further themselves, to one workgroup per CU)


def _stream_key(dev):
    """One scratch / workspace per device for the launches of the caller's stream (a captured graph replays on it and uses the
    buffers the eager first step created), a second one for the opt-in side stream (its launches run beside the main stream's)."""
    d = torch.device(dev)
    idx = d.index if d.index is not None else torch.cuda.current_device()
    side = _side_stream is not None and torch.cuda.current_stream(idx).cuda_stream == _side_stream.cuda_stream
    return idx, bool(side)


_retired: list = []          # outgrown scratch buffers: kept alive, a graph captured earlier may still write into them


def _pad4(n: int) -> int:
    return (n + 3) // 4 * 4


def _scratch_slabs(dev, n_elems: int, cap: int):
    """(slab tensor, stride) for a result of n_elems elements that is summed right away on the current stream"""
    stride = _pad4(n_elems)
    key = _stream_key(dev)
    buf = _slab_scratch.get(key)
    if buf is None or buf.numel() < cap * stride:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("k-split scratch would have to be allocated during a graph capture: run one eager step first")
        if buf is not None:
            _retired.append(buf)
        buf = _slab_scratch[key] = torch.empty(max(cap * stride, 1 << 24), device=dev, dtype=torch.float32)
    return buf, stride


def _slab_store(owner) -> dict:
    return _slab_param if owner is None else owner.__dict__.setdefault("_slab_store", {})


def _param_slabs(c: torch.Tensor, cap: int, owner=None):
    """(slab tensor, stride) that lives as long as the optimiser that owns the gradient view `c`"""
    stride = _pad4(c.numel())
    key = (c.data_ptr(), c.numel(), cap)
    store = _slab_store(owner)
    buf = store.get(key)
    if buf is None:
        buf = store[key] = torch.empty(cap * stride, device=c.device, dtype=torch.float32)
    return buf, stride


def _fold_entry(c_ptr: int, slab_ptr: int, stride: int, n: int, nslab: int):
    d = _lib.SlabDesc()
    d.c, d.slab, d.slab_stride, d.n, d.nslab = c_ptr, slab_ptr, stride, n, nslab
    return d


def _fold_launch(entries):
    if entries:
        arr = (_lib.SlabDesc * len(entries))(*entries)
        check(lib().dvae_slab_fold(arr, len(entries), stream()), "dvae_slab_fold")


def _defer_fold(grad: torch.Tensor, owner, slab_ptr: int, stride: int, nslab: int, n: int = None):
    """`nslab` slabs are to be added to `grad` (n elements, a multiple of 4): later when an optimiser owns it, now otherwise"""
    if nslab < 1:
        return
    n = _pad4(grad.numel()) if n is None else n
    e = _fold_entry(grad.data_ptr(), slab_ptr, stride, n, nslab)
    if owner is None:
        _fold_launch([e])
        return
    if not _queue_end_of_backward():      # not inside a backward pass (an op's backward called by hand): sum right away
        _fold_launch([e])
        return
    pend = owner.__dict__.setdefault("_slab_pending", {})
    key = (grad.data_ptr(), slab_ptr)
    if key in pend:            # a second contribution to this gradient before the first was summed: sum that one now
        _fold_launch([pend.pop(key)])
    pend[key] = e
    if not any(o is owner for o in _fold_owners):
        _fold_owners.append(owner)


def fold_pending(owner=None, params=None):
    """Add every pending k-split slab to its gradient (one launch per 64 results): `owner`'s, or those of `params` only."""
    if params is not None:
        for p in params:
            ow = getattr(p, "_dvae_owner", None)
            pend = getattr(ow, "_slab_pending", None) if ow is not None else None
            if pend and p.grad is not None:
                lo, hi = p.grad.data_ptr(), p.grad.data_ptr() + 4 * p.grad.numel()
                ks = [k for k in pend if lo <= k[0] < hi]
                _fold_launch([pend.pop(k) for k in ks])
        return
    pend = getattr(owner, "_slab_pending", None)
    if pend:
        _fold_launch(list(pend.values()))
        pend.clear()


def wgrad_gemm_batched(As, Bs, params, M, N, K, lda, ldb, split_k, mode, flags=0):
    """the weight gradients of len(params) (<= 4) parameters of one shape in ONE launch (dvae_gemm_f32_batched_slabs): each
    += its product, k-splits behind the first into slabs (see wgrad_gemm)"""
    a = lambda t: t if isinstance(t, int) else t.data_ptr()
    nb = len(params)
    grads = [_grad_buf(p) for p in params]
    arr = lambda xs: (C.c_void_p * nb)(*[a(x) for x in xs])
    m = _mflags(_mode(mode), As[0], Bs[0], grads[0]) | flags
    if split_k <= 1:
        check(lib().dvae_gemm_f32_batched(arr(As), arr(Bs), arr(grads), nb, M, N, K, lda, ldb, N, 0, 0, EPI_ACCUM, 1, m,
                                          stream()), "dvae_gemm_f32_batched")
        return
    owner = _owner_of(params[0])
    stride = _pad4(M * N)
    cap = _cap_for(split_k) * nb
    key = (grads[0].data_ptr(), M * N, cap)
    slab = _slab_store(owner).get(key) if owner is not None else None
    if slab is None:
        if owner is not None:
            slab = _slab_store(owner)[key] = torch.empty(cap * stride, device=grads[0].device, dtype=torch.float32)
        else:
            slab, _ = _scratch_slabs(grads[0].device, cap * stride, 1)
    n = lib().dvae_gemm_f32_batched_slabs(arr(As), arr(Bs), arr(grads), nb, ptr(slab), stride, cap, M, N, K, lda, ldb, N, 0, 0,
                                          EPI_ACCUM, split_k, m, stream())
    if n < 1:
        check(n, "dvae_gemm_f32_batched_slabs")
    if n > 1:
        for b, (g, p) in enumerate(zip(grads, params)):
            _defer_fold(g, _owner_of(p), slab.data_ptr() + 4 * b * n * stride, stride, n)


def _owner_of(p):
    return getattr(p, "_dvae_owner", None)


def gemm_slabs(A, B, Cout, bias, M, N, K, lda, ldb, ldc, a_kc, b_kc, epi, split_k, mode, slab, stride, cap, flags=0):
    """dvae_gemm_f32_slabs: returns the number n of k-splits launched: n > 1 -> every split in its slab, Cout untouched;
    n == 1 -> Cout written as `epi` says"""
    a = lambda t: t if (t is None or isinstance(t, int)) else t.data_ptr()
    n = lib().dvae_gemm_f32_slabs(a(A), a(B), a(Cout), a(slab), stride, cap, a(bias), M, N, K, lda, ldb, ldc, int(a_kc),
                                  int(b_kc), epi, split_k, _mflags(_mode(mode), A, B, Cout) | flags, stream())
    if n < 1:
        check(n, "dvae_gemm_f32_slabs")
    return n


def _cap_for(split_k: int) -> int:
    """slabs to provide for a product asked to split `split_k` ways: the one-workgroup-per-CU tiles may double the split"""
    return max(SLAB_CAP, min(128, 2 * int(split_k)))


def gemm_split(A, B, Cout, bias, M, N, K, lda, ldb, ldc, a_kc, b_kc, act, split_k, mode, flags=0):
    """Cout = act(A B + bias) with K cut into (about) split_k parts and NO atomics: partial products into scratch slabs,
    summed (and activated) right behind the contraction.  Cout: [M, N] contiguous (ldc == N)."""
    n_el = M * ldc
    cap = _cap_for(split_k)
    slab, stride = _scratch_slabs(Cout.device, n_el, cap)
    n = gemm_slabs(A, B, Cout, bias, M, N, K, lda, ldb, ldc, a_kc, b_kc, EPI_STORE, split_k, mode, slab, stride, cap, flags)
    if n > 1:       # Cout = act(sum of the n slabs) (the bias rode split 0)
        check(lib().dvae_slab_sum(ptr(Cout), ptr(slab), stride, n, n_el, act, 0, stream()), "dvae_slab_sum")
    elif act != ACT_NONE:
        check(lib().dvae_slab_sum(ptr(Cout), None, 0, 0, n_el, act, 1, stream()), "dvae_slab_sum")


def wgrad_gemm(A, B, grad, owner, M, N, K, lda, ldb, a_kc, b_kc, split_k, mode, flags=0):
    """grad[M, N] += A^T B-shaped weight gradient, K cut into (about) split_k parts without atomics: every split STORES its slab
    (no read-modify-write in any epilogue); `grad += their sum` later (optimiser-owned gradient: end of backward) or now."""
    if split_k <= 1:
        gemm(A, B, grad, None, M, N, K, lda, ldb, N, a_kc, b_kc, ACT_NONE, EPI_ACCUM, 1, mode, flags)
        return
    cap = _cap_for(split_k)
    slab, stride = _param_slabs(grad, cap, owner) if owner is not None else _scratch_slabs(grad.device, grad.numel(), cap)
    n = gemm_slabs(A, B, grad, None, M, N, K, lda, ldb, N, a_kc, b_kc, EPI_ACCUM, split_k, mode, slab, stride, cap, flags)
    if n > 1:
        _defer_fold(grad, owner, slab.data_ptr(), stride, n)


_DETERMINISTIC = False


def set_deterministic(flag: bool = True):
    """TEST mode (not the benchmarked path): every contraction runs UNSPLIT.  Since round 6 the default mode is run-to-run
    bit-identical as well (k-splits store slabs that are added in a fixed order, deterministic column sums and recurrence
    bias gradients: no floating-point atomics on the step); this mode is a second deterministic arithmetic with another
    summation order — tests/test_hip_determinism.py runs every comparison in both.  Slower (under-filled launches);
    `DVAE_DETERMINISTIC=1` in the environment switches it on at import."""
    global _DETERMINISTIC
    _DETERMINISTIC = bool(flag)
    check(lib().dvae_set_deterministic(int(_DETERMINISTIC)), "dvae_set_deterministic")


def deterministic() -> bool:
    return _DETERMINISTIC


def _split_k(m_tiles: int, k: int, slots: int = 0, fixed: int = 192) -> int:
    """Pick the split of the contraction that minimises (rounds over the chip) x (k per workgroup + fixed cost):
    512 workgroup slots (2 per CU at the 128x128x32 tile), each split keeps >= 256 of K.  Not restricted to
    powers of two: 80 tiles x 6 splits fill one round where x 8 needs two.  `slots` / `fixed`: for shapes that run on the
    256 x 128 kernels (one workgroup per CU; a 128 KB epilogue per workgroup — a slab store since round 6; the constants were
    fitted to the atomic epilogues of rounds 3-5 and re-checked against the same-box A/B of round 6)."""
    if _DETERMINISTIC:
        return 1
    slots = slots or int(os.environ.get("DVAE_SPLIT_SLOTS", "512"))
    best, best_cost = 1, None
    for s in range(1, max(1, k // 256) + 1):
        rounds = -(-(m_tiles * s) // slots)
        # + 4 per split: each split is one more epilogue over the output, and one more slab for the fold (dW[256 x 64] over K = 65536: 128 splits
        # 36 us, the 256 a split-free model picks 46; scripts/skinny_wgrad_sweep.py)
        cost = rounds * (-(-k // s) + fixed) * (1.0 if s == 1 else 1.03) + (4 * s if s > 1 else 0)
        if best_cost is None or cost < best_cost:
            best, best_cost = s, cost
    return best


def _tiles(m, n):
    return ((m + 127) // 128) * ((n + 127) // 128)


# ----------------------------------------------------------------------------- compute mode
MODE_F32, MODE_BF16, MODE_F32X3 = _lib.MODE_F32, _lib.MODE_BF16, _lib.MODE_F32X3
A_BF16, B_BF16, C_BF16 = 0x100, 0x200, 0x400    # DVAE_MODE_A/B/C_BF16: operand / result stored as bf16 (bf16 mode)
DEFAULT_COMPUTE_DTYPE = _lib.DEFAULT_COMPUTE_DTYPE      # "fp32x3" (DVAE_COMPUTE_DTYPE in the environment overrides)
_MODE_NAMES = {MODE_F32: "fp32", MODE_BF16: "bf16", MODE_F32X3: "fp32x3"}


def set_compute_dtype(name: str):
    """Arithmetic of every contraction (Linear, Conv1d, LSTM input projection and recurrence, all weight gradients):
    "fp32"    fp32 operands on the fp32 MFMA (v_mfma_f32_32x32x2_f32 / 16x16x4_f32; runs at the vector rate);
    "fp32x3"  (default) fp32 RESULTS on the bf16 matrix pipe: operands split exactly into three bf16 terms, six exact
              partial products per product, fp32 accumulation (include/dvae_hip.h, DVAE_MODE_F32X3) — BASELINE
              configs[1]/[3];
    "bf16"    operands rounded to bf16, fp32 accumulation — BASELINE configs[2]/[4].
    Tensors in HBM, BatchNorm, gates, losses, master weights and Adam are fp32 in every mode.  The mode in force when
    an op's FORWARD runs is recorded in its autograd context and used by its backward launches too."""
    if name not in _lib.COMPUTE_MODES:
        raise ValueError(f"compute dtype {name!r}: expected one of {sorted(_lib.COMPUTE_MODES)}")
    check(lib().dvae_set_compute_mode(_lib.COMPUTE_MODES[name]), "dvae_set_compute_mode")


def current_mode() -> int:
    return lib().dvae_get_compute_mode()


def get_compute_dtype() -> str:
    return _MODE_NAMES[current_mode()]


def _mflags(mode, A=None, B=None, Cout=None):
    """`mode` plus the storage flags of bf16 tensors (bf16 compute mode only; anything else must be fp32)."""
    m = int(mode)
    bf = lambda t: t is not None and not isinstance(t, int) and t.dtype == torch.bfloat16
    if bf(A) or bf(B) or bf(Cout):
        if (m & 0xff) != MODE_BF16:
            raise ValueError("bf16 tensors are operands of the bf16 compute mode only")
        m |= (A_BF16 if bf(A) else 0) | (B_BF16 if bf(B) else 0) | (C_BF16 if bf(Cout) else 0)
    return m


def _mode(mode):
    if mode is None:
        return current_mode()
    return _lib.COMPUTE_MODES[mode] if isinstance(mode, str) else int(mode)


class compute_dtype:
    """Context manager: `with ops.compute_dtype("bf16"): ...`"""

    def __init__(self, name):
        self.name = name

    def __enter__(self):
        self.prev = get_compute_dtype()
        set_compute_dtype(self.name)

    def __exit__(self, *exc):
        set_compute_dtype(self.prev)


# ----------------------------------------------------------------------------- raw launches
def gemm(A, B, Cout, bias, M, N, K, lda, ldb, ldc, a_kc, b_kc, act=ACT_NONE, epi=EPI_STORE, split_k=1, mode=None,
         flags=0):
    """A, B, Cout, bias: tensors or raw device addresses (ints).  Storage flags (A_BF16 ...) are taken from the dtypes of
    tensor arguments; for raw addresses pass them in `flags`."""
    a = lambda t: t if (t is None or isinstance(t, int)) else t.data_ptr()
    check(lib().dvae_gemm_f32(a(A), a(B), a(Cout), a(bias), M, N, K, lda, ldb, ldc, int(a_kc), int(b_kc),
                              act, epi, split_k, _mflags(_mode(mode), A, B, Cout) | flags, stream()), "dvae_gemm_f32")


def gemm_batched(As, Bs, Cs, M, N, K, lda, ldb, ldc, a_kc, b_kc, epi=EPI_ATOMIC, split_k=1, mode=None, flags=0):
    """len(As) (<= 4) products of one shape in ONE launch (dvae_gemm_f32_batched): C[b] (+)= opA(A[b]) opB(B[b]).
    As / Bs / Cs: tensors or raw device addresses."""
    a = lambda t: t if isinstance(t, int) else t.data_ptr()
    n = len(As)
    arr = lambda xs: (C.c_void_p * n)(*[a(x) for x in xs])
    m = _mode(mode)
    check(lib().dvae_gemm_f32_batched(arr(As), arr(Bs), arr(Cs), n, M, N, K, lda, ldb, ldc, int(a_kc), int(b_kc), epi,
                                      split_k, _mflags(m, As[0], Bs[0], Cs[0]) | flags, stream()), "dvae_gemm_f32_batched")


def linear_fwd(x, w, b, act=ACT_NONE, mode=None, w16=None):
    """y[M,Nout] = act(x[M,K] @ w[Nout,K]^T + b).  w16: bf16 copy of w (bf16 compute mode), used as the operand."""
    M, K = x.shape
    w = w if w16 is None else w16
    Nout = w.shape[0]
    sk = _split_k(_tiles(M, Nout), K)
    y = torch.empty((M, Nout), device=x.device, dtype=torch.float32)
    if sk > 1 and (M * Nout) % 4 == 0:
        gemm_split(x, w, y, b, M, Nout, K, K, K, Nout, True, True, act, sk, mode)
    else:
        gemm(x, w, y, b, M, Nout, K, K, K, Nout, True, True, act, EPI_STORE, 1, mode)
    return y


def linear_dgrad(dy, w, mode=None, w16=None):
    """dx[M,K] = dy[M,Nout] @ w[Nout,K]"""
    w = w if w16 is None else w16
    M, Nout = dy.shape
    K = w.shape[1]
    sk = _split_k(_tiles(M, K), Nout)
    dx = torch.empty((M, K), device=dy.device, dtype=torch.float32)
    if sk > 1 and (M * K) % 4 == 0:
        gemm_split(dy, w, dx, None, M, K, Nout, Nout, K, K, True, False, ACT_NONE, sk, mode)
    else:
        gemm(dy, w, dx, None, M, K, Nout, Nout, K, K, True, False, mode=mode)
    return dx


def linear_wgrad_acc(dy, x, wgrad, rows=None, lda=None, ldb=None, mode=None, store=False, owner=None):
    """wgrad[Nout,K] += dy[rows,Nout]^T @ x[rows,K] (split over rows into slabs, summed later when `owner` — the optimiser
    that owns the gradient buffer — is given, right away otherwise).
    store: the caller guarantees this launch is the ONLY contribution to `wgrad` this step and that nothing zeroed it
    (FlatAdam.zero_grad skips such parameters): an unsplit launch then STORES instead of read-modify-writing — for the two
    134 MB outer products (enc_linear, dec_pre_linear2: K = 2B rows only) that halves an HBM-bound launch."""
    Nout, K = wgrad.shape
    rows = dy.shape[0] if rows is None else rows
    lda = Nout if lda is None else lda
    ldb = K if ldb is None else ldb
    sk = _split_k(_tiles(Nout, K), rows)
    if store:
        if sk != 1:
            raise ValueError("store-first weight gradient needs an unsplit launch")
        gemm(dy, x, wgrad, None, Nout, K, rows, lda, ldb, K, False, False, ACT_NONE, EPI_STORE, 1, mode)
        return
    # one split: a plain read-modify-write (coalesced 128-B rows); several: split 0 the same, the others into slabs — no atomics
    wgrad_gemm(dy, x, wgrad, owner, Nout, K, rows, lda, ldb, False, False, sk, mode)


def colsum_add(x, out1, out2=None, rows=None, cols=None, ld=None):
    rows = x.shape[0] if rows is None else rows
    cols = x.shape[1] if cols is None else cols
    ld = x.shape[1] if ld is None else ld
    key = _stream_key(x.device)
    need = lib().dvae_colsum_ws_bytes(rows, cols)
    ws = _colsum_ws.get(key)
    if ws is None or ws.numel() < need:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("column-sum workspace would have to be allocated during a graph capture: run one eager step first")
        if ws is not None:
            _retired.append(ws)
        ws = _colsum_ws[key] = torch.zeros(max(need, 1 << 23), device=x.device, dtype=torch.uint8)   # counters start at zero
    check(lib().dvae_colsum_add_ws(ptr(x), ptr(out1), ptr(out2), rows, cols, ld, _b16(x), ptr(ws), stream()),
          "dvae_colsum_add_ws")


def mel_to_frames(x1, x2=None, dtype=torch.float32):
    """[Bh,C,T] (x2 optional) -> frame-major [T*N, C] (fp32, or bf16 when it only feeds a contraction in the bf16 mode)"""
    _ok(x1, x2)
    Bh, Cc, T = x1.shape
    N = Bh * (2 if x2 is not None else 1)
    X = torch.empty((T * N, Cc), device=x1.device, dtype=dtype)
    check(lib().dvae_mel_to_frames(ptr(x1), ptr(x2), ptr(X), Bh, Cc, T, _b16(X), stream()), "dvae_mel_to_frames")
    return X


def transpose2d(x):
    R, Cc = x.shape
    out = torch.empty((Cc, R), device=x.device, dtype=torch.float32)
    check(lib().dvae_transpose(ptr(x), ptr(out), R, Cc, stream()), "dvae_transpose")
    return out


# ----------------------------------------------------------------------------- Linear (+ReLU)
class LinearFn(torch.autograd.Function):
    """nn.Linear (+ optional ReLU): disentangled_vae.py:165-171,194 used at :211-213,232-233,247."""

    @staticmethod
    def forward(ctx, x, weight, bias, act, w16=None, x16=None):
        """x16: bf16 data of x when x is a placeholder (output of an LSTM layer in the bf16 compute mode)."""
        x = x if x16 is None else x16
        _ok(x, weight, bias, act16=(x,))
        ctx.mode = current_mode()
        ctx.w16 = w16 if ctx.mode == MODE_BF16 else None      # bf16 copy of the weight (derived.DerivedWeights)
        y = linear_fwd(x, weight, bias, act, ctx.mode, ctx.w16)
        ctx.save_for_backward(x, weight, bias, y if act != ACT_NONE else None)
        ctx.act = act
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, bias, y = ctx.saved_tensors
        dy = dy.contiguous()
        if ctx.act != ACT_NONE:
            du = torch.empty_like(dy)
            check(lib().dvae_act_bwd(ptr(dy), ptr(y), ptr(du), dy.numel(), ctx.act, stream()), "dvae_act_bwd")
            dy = du
        dx = linear_dgrad(dy, weight, ctx.mode, ctx.w16) if ctx.needs_input_grad[0] else None
        with side_work(dy, x):
            # a parameter FlatAdam.zero_grad leaves alone because exactly one unsplit launch writes its gradient per step
            store = bool(getattr(weight, "_dvae_grad_store_first", False)) and \
                _split_k(_tiles(weight.shape[0], weight.shape[1]), dy.shape[0]) == 1
            if getattr(weight, "_dvae_grad_store_first", False) and not store:
                raise RuntimeError("store-first weight gradient: the launch would be split; clear the flag "
                                   "(FlatAdam.set_store_first) for this shape")
            linear_wgrad_acc(dy, x, _grad_buf(weight), mode=ctx.mode, store=store, owner=_owner_of(weight))
            if store:
                weight._dvae_sf_writes = getattr(weight, "_dvae_sf_writes", 0) + 1     # FlatAdam.step checks: exactly one
            colsum_add(dy, _grad_buf(bias))
        _ready(weight, bias)
        return dx, None, None, None, None, None


# ----------------------------------------------------------------------------- Conv1d(k5) + BatchNorm + act
def _placeholder(R, C, dev):
    """fp32 [R, C] tensor with one element of storage: what autograd sees of an activation whose DATA lives in a bf16
    tensor travelling beside it (bf16 compute mode).  Gradients flow through the placeholder in fp32 — autograd would
    otherwise cast every gradient of a bf16 tensor to bf16 — nobody reads its values."""
    return torch.empty_strided((R, C), (0, 0), device=dev, dtype=torch.float32)


class ConvBnActFn(torch.autograd.Function):
    """act(BatchNorm1d_train(Conv1d_k5_p2(x))) [+ residual] on frame-major rows.
    Reference blocks: encoder :151-162/:201-202, decoder :175-191/:242-243, Postnet :43-87.
    `conv_wp` is the conv weight in the PACKED layout [5][Cout][Cin] (the layout the parameter lives in, see
    model/disentangled_vae._Conv1dParams); its gradient is accumulated in the same layout.  `wpt` is the transposed
    pack [5][Cin][Cout] for the data gradient (derived.DerivedWeights) or None (computed in backward if needed).
    bf16 compute mode: `w16` / a bf16 `wpt` are bf16 copies of the weights; `x16` is the bf16 DATA of the input when `x`
    is a placeholder (see _placeholder); with `emit16` the op returns (z, z16): z16 the bf16 output written by the
    BatchNorm-apply kernel (the storage its consumers — contractions — read), z its fp32 placeholder for autograd; in the
    other modes, with a residual, or in eval mode z is a real fp32 tensor and z16 is None."""

    @staticmethod
    def forward(ctx, x, conv_wp, conv_b, bn_w, bn_b, running_mean, running_var, nbt, residual,
                n_seg, groups, act, training, wpt=None, w16=None, x16=None, emit16=False):
        _ok(x if x16 is None else x16, conv_wp, conv_b, bn_w, bn_b, residual, act16=(x, x16))
        L = lib()
        R, Cin = x.shape
        if conv_wp.dim() != 3 or conv_wp.shape[0] != 5 or conv_wp.shape[2] != Cin:
            raise ValueError(f"conv weight must be packed [5, Cout, Cin={Cin}], got {tuple(conv_wp.shape)}")
        Cout = conv_wp.shape[1]
        dev = x.device
        st = stream()
        mode = current_mode()
        xa = x if x16 is None else x16                                        # what the kernels read
        wop = w16 if (w16 is not None and mode == MODE_BF16) else conv_wp     # bf16 copy of the pack in the bf16 mode
        fmode = _mflags(mode, xa, wop)
        y = torch.empty((R, Cout), device=dev, dtype=torch.float32)
        if training:
            # the conv epilogue leaves the BatchNorm partial sums of y behind: no separate statistics pass over y
            G = groups
            mean = torch.empty((G, Cout), device=dev, dtype=torch.float32)
            rstd = torch.empty((G, Cout), device=dev, dtype=torch.float32)
            ws = torch.empty((L.dvae_bn_ws_bytes(R, Cout, G),), device=dev, dtype=torch.uint8)
            if 64 < Cout <= 128 and R >= 4096 and mode == MODE_F32X3 and (R * Cout) % 4 == 0:
                # few output columns (postnet's last conv: 80 mel channels): the conv is cut along k (csrc/gemm.hip
                # narrow_conv_split), its splits stored into slabs and summed right here (no atomics) — no statistics in
                # that epilogue: one pass over the [R, 80] result instead
                slab, stride = _scratch_slabs(dev, R * Cout, 8)
                n = L.dvae_conv5_fwd_slabs(ptr(xa), ptr(wop), ptr(conv_b), ptr(y), ptr(slab), stride, 8, R, n_seg, Cin, Cout,
                                           fmode, st)
                if n < 1:
                    check(n, "dvae_conv5_fwd_slabs")
                if n > 1:
                    check(L.dvae_slab_sum(ptr(y), ptr(slab), stride, n, R * Cout, ACT_NONE, 0, st), "dvae_slab_sum")
                check(L.dvae_bn_stats_fwd(ptr(y), ptr(mean), ptr(rstd), ptr(running_mean), ptr(running_var), ptr(nbt),
                                          ptr(ws), R, n_seg, Cout, G, BN_EPS, BN_MOMENTUM, st), "dvae_bn_stats_fwd")
            else:
                check(L.dvae_conv5_fwd_stats(ptr(xa), ptr(wop), ptr(conv_b), ptr(y), R, n_seg, Cin, Cout, fmode, G,
                                             ptr(ws), st), "dvae_conv5_fwd_stats")
                check(L.dvae_bn_stats_finalize(ptr(mean), ptr(rstd), ptr(running_mean), ptr(running_var), ptr(nbt),
                                               ptr(ws), R, n_seg, Cout, G, BN_EPS, BN_MOMENTUM, st), "dvae_bn_stats_finalize")
        else:
            check(L.dvae_conv5_fwd(ptr(xa), ptr(wop), ptr(conv_b), ptr(y), R, n_seg, Cin, Cout, fmode, st),
                  "dvae_conv5_fwd")
            G = 1
            mean = running_mean.detach().reshape(1, Cout).contiguous()
            rstd = torch.rsqrt(running_var.detach() + BN_EPS).reshape(1, Cout).contiguous()
        as16 = bool(emit16) and mode == MODE_BF16 and training and residual is None
        zd = torch.empty((R, Cout), device=dev, dtype=torch.bfloat16 if as16 else torch.float32)   # the data
        check(L.dvae_bn_apply_fwd(ptr(y), ptr(mean), ptr(rstd), ptr(bn_w), ptr(bn_b), ptr(residual), ptr(zd),
                                  R, n_seg, Cout, G, act, _b16(zd), st), "dvae_bn_apply_fwd")
        z = _placeholder(R, Cout, dev) if as16 else zd
        ctx.save_for_backward(xa, y, zd, mean, rstd, conv_wp, conv_b, bn_w, bn_b, wpt)
        ctx.cfg = (n_seg, G, act, training, residual is not None, mode, Cin)
        ctx.n_out = 2 if emit16 else 1
        if not emit16:
            return z
        if as16:
            ctx.mark_non_differentiable(zd)
            ctx.set_materialize_grads(False)      # no zero-fill for the bf16 twin's (never used) gradient
        return z, (zd if as16 else None)

    @staticmethod
    def backward(ctx, dz, *_):
        xa, y, z, mean, rstd, conv_wp, conv_b, bn_w, bn_b, wpt = ctx.saved_tensors
        n_seg, G, act, training, has_res, mode, Cin = ctx.cfg
        if not training:
            raise RuntimeError("backward through eval-mode BatchNorm is not part of the training path")
        L = lib()
        st = stream()
        dz = dz.contiguous()
        R = y.shape[0]
        Cout = y.shape[1]
        dev = y.device
        if has_res and act != ACT_NONE:
            raise RuntimeError("residual is only supported with ACT_NONE")
        ws = torch.empty((L.dvae_bn_ws_bytes(R, Cout, G),), device=dev, dtype=torch.uint8)
        dy = torch.empty((R, Cout), device=dev, dtype=act_storage(mode))     # operand of the data / weight gradient
        if act in (ACT_RELU, ACT_NONE):
            # the mask of a ReLU block is recomputed from Y: the pass does not read Z
            check(L.dvae_bn_bwd_from_y(ptr(dz), ptr(y), ptr(mean), ptr(rstd), ptr(bn_w), ptr(bn_b), ptr(dy),
                                       ptr(_grad_buf(bn_w)), ptr(_grad_buf(bn_b)), ptr(ws), R, n_seg, Cout, G, act,
                                       _b16(dy) << 1, st), "dvae_bn_bwd_from_y")
        else:
            check(L.dvae_bn_bwd(ptr(dz), ptr(y), ptr(z), ptr(mean), ptr(rstd), ptr(bn_w), ptr(dy),
                                ptr(_grad_buf(bn_w)), ptr(_grad_buf(bn_b)), ptr(ws), R, n_seg, Cout, G, act,
                                _b16(z) | (_b16(dy) << 1), st), "dvae_bn_bwd")
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty((R, Cin), device=dev, dtype=torch.float32)
            # the data gradient contracts over Cout: against the transposed pack [5][Cin][Cout] both operands are
            # k-contiguous (the ds_read_b128 fragment path)
            if wpt is None:
                from .derived import conv_wpt_local
                wpt = conv_wpt_local(conv_wp)
            if Cin <= 128 and (R * Cin) % 4 == 0:
                # few output columns (the 80 mel channels): the default arithmetic cuts K to fill the chip — into slabs
                slab, stride = _scratch_slabs(dev, R * Cin, 8)
                n = L.dvae_conv5_dgrad_t_slabs(ptr(dy), ptr(wpt), ptr(dx), ptr(slab), stride, 8, R, n_seg, Cin, Cout,
                                               _mflags(mode, dy, wpt), st)
                if n < 1:
                    check(n, "dvae_conv5_dgrad_t_slabs")
                if n > 1:
                    check(L.dvae_slab_sum(ptr(dx), ptr(slab), stride, n, R * Cin, ACT_NONE, 0, st), "dvae_slab_sum")
            else:
                check(L.dvae_conv5_dgrad_t(ptr(dy), ptr(wpt), ptr(dx), R, n_seg, Cin, Cout, _mflags(mode, dy, wpt), st),
                      "dvae_conv5_dgrad_t")
        with side_work(dy, xa):
            # the weight gradient goes straight into the (packed) gradient view: atomic split-K epilogue
            if Cout >= 256 and Cin > 64:
                # the 256 x 128 kernels: 256 slots, and every split costs each output tile another atomic epilogue.
                # bf16, R = 65536, 512 -> 512: 6 splits 162 us, the 19 the 128-tile model picks 210 (scripts/wgrad_split_sweep.py)
                sk = _split_k(5 * ((Cout + 255) // 256) * ((Cin + 127) // 128), R, slots=256, fixed=384)
            else:
                sk = _split_k(5 * _tiles(Cout, Cin), R)
            # split 0 adds to the (packed) gradient view, the other k-splits store slabs that are summed in front of the
            # Adam launch (or right here for a gradient no optimiser owns): no atomics
            gw, own = _grad_buf(conv_wp), _owner_of(conv_wp)
            slab, stride = (None, 0)
            cap = _cap_for(sk)
            if sk > 1:
                slab, stride = _param_slabs(gw, cap, own) if own is not None else _scratch_slabs(dev, gw.numel(), cap)
            n = L.dvae_conv5_wgrad_slabs(ptr(dy), ptr(xa), ptr(gw), ptr(slab), stride, cap if sk > 1 else 0, R, n_seg,
                                         Cin, Cout, EPI_ACCUM, sk, _mflags(mode, dy, xa), stream())
            if n < 1:
                check(n, "dvae_conv5_wgrad_slabs")
            if n > 1:
                _defer_fold(gw, own, slab.data_ptr(), stride, n)
            colsum_add(dy, _grad_buf(conv_b))
        _ready(conv_wp, conv_b, bn_w, bn_b)
        dres = dz if has_res else None
        return (dx, None, None, None, None, None, None, None, dres, None, None, None, None, None, None, None, None)


# ----------------------------------------------------------------------------- LSTM layer
# W_hh-resident persistent recurrence (csrc/lstm_pers.hip): one launch per sequence where the shape has such a kernel
# (bf16 compute mode, H = 512 / 1024, (H/32) * ceil(N/32) <= CU count); DVAE_LSTM_PERSISTENT=0 keeps one launch per frame.
LSTM_PERSISTENT = os.environ.get("DVAE_LSTM_PERSISTENT", "1") != "0"


def _ranks_share_a_gpu() -> bool:
    """More ranks on this NODE than visible GPUs (a functional check, never the product set-up): two persistent grids
    each want every CU of the device, neither becomes resident, both run into their bounded waits.  Decided from
    LOCAL_WORLD_SIZE only (torch.distributed.run and this repository's launchers export it): WORLD_SIZE counts the ranks of
    every node, and a multi-node job must not lose its persistent recurrences to it."""
    try:
        if "LOCAL_WORLD_SIZE" not in os.environ:
            return False
        n_local = int(os.environ["LOCAL_WORLD_SIZE"])
        n_dev = torch.cuda.device_count()          # does not initialise HIP
        return n_dev > 0 and n_local > n_dev
    except Exception:
        return False


if _ranks_share_a_gpu():
    import warnings
    warnings.warn(f"dvae_amd: LOCAL_WORLD_SIZE={os.environ.get('LOCAL_WORLD_SIZE')} ranks share {torch.cuda.device_count()} "
                  "visible GPU(s): the W_hh-resident persistent recurrences are switched OFF (one launch per frame, much "
                  "slower) — a functional check, not the product set-up")
    LSTM_PERSISTENT = False
LSTM_PERS_TIMEOUT_US = 0                    # 0: the library's default bound (2 s) on every cross-workgroup wait
_PERS_WS_BYTES = (1 << 20) + 4 * 16 * 128 * 2 * 1024     # >= the workspace any supported (N, H) needs (four ring slots)
_pers_ws: dict = {}
_pers_owner: dict = {}                      # device index -> the torch stream whose launches used the workspace last


def lstm_persistent_usable(N: int, H: int, mode: int, ndir: int = 1, bwd: bool = False) -> bool:
    """`mode`: precision of the recurrent product of that pass (derived.lstm_pack_modes)."""
    return bool(LSTM_PERSISTENT and ndir == 1 and lib().dvae_lstm_pers_supported(N, H, int(mode), int(bwd))
                and lib().dvae_lstm_pers_ws_bytes(N, H) <= _PERS_WS_BYTES)


def lstm_pers_workspace(dev) -> torch.Tensor:
    """Flags + sticky error record + the flags' epoch + exchange ring of the persistent LSTM launches of one device: zeroed
    ONCE, here, and never touched by the host again — a flag holds `epoch + frames published`, every launch reads the epoch
    its predecessor left and the last workgroup to finish advances it (csrc/lstm_pers.hip pers_epoch), so no launch has to
    clear anything in front of it; launches of one stream share the workspace.
    (One exchange slot PER FRAME read with plain, L2-cached loads was tried and removed: wrong results under hipGraph
    replay — blamed on stale L2 lines then, possibly the flag-clear hazard of the memset node that was in use — and no
    faster; DESIGN_HISTORY.md.)"""
    key = torch.device(dev).index if torch.device(dev).index is not None else torch.cuda.current_device()
    ws = _pers_ws.get(key)
    if ws is None:
        ws = _pers_ws[key] = torch.zeros(_PERS_WS_BYTES, device=f"cuda:{key}", dtype=torch.uint8)
        assert ws.data_ptr() % 256 == 0
    return ws


def lstm_pers_local_launches(dev="cuda") -> int:
    """Statistics word of the workspace: the persistent launches so far whose cross-workgroup hand-offs stayed in their XCD's
    L2 (csrc/lstm_pers.hip pers_loc_*: geometries with a multiple of 8 row groups, after the workgroups have verified at frame 1
    that each row group shares one XCD; DVAE_PERS_XCD_LOCAL=0 switches the form off).  Synchronises the device."""
    ws = lstm_pers_workspace(dev)
    torch.cuda.synchronize(ws.device)
    return int(ws[65536 + 40:65536 + 44].view(torch.int32).item())      # PERS_ERR_OFF + 4 * PERS_LOCAL_WORD


def _pers_claim(dev):
    """ONE persistent launch at a time per device: a launch owns every CU and the device's one workspace (flags, exchange
    ring).  Launches of one stream are ordered by the stream; a DIFFERENT stream may take over only when the previous
    owner has nothing in flight — otherwise two grids would clear and overwrite each other's flags and ring (silent
    corruption of h / dG, not a timeout) and neither could become fully resident.  Raises instead.  A capturing stream
    is exempt (torch synchronises the device around a capture; a replayed graph runs on the stream that replays it)."""
    if torch.cuda.is_current_stream_capturing():
        return
    key = torch.device(dev).index if torch.device(dev).index is not None else torch.cuda.current_device()
    cur = torch.cuda.current_stream(key)
    own = _pers_owner.get(key)
    if own is not None and own.cuda_stream != cur.cuda_stream and not own.query():
        raise RuntimeError("persistent LSTM launch requested on a second stream while launches of another stream are "
                           "still in flight on this device: one persistent recurrence at a time per GPU (synchronise the "
                           "streams, or set DVAE_LSTM_PERSISTENT=0 for concurrent trainers)")
    _pers_owner[key] = cur


def _pers_fill(d, dev):
    """pers_ws / pers_timeout_us of a dvae_lstm_dir_t"""
    _pers_claim(dev)
    d.pers_ws, d.pers_timeout_us = ptr(lstm_pers_workspace(dev)), LSTM_PERS_TIMEOUT_US


def lstm_pers_check():
    """Synchronises the current stream of every device that ran a persistent LSTM launch and raises if a bounded
    cross-workgroup wait gave up (the launch's outputs are garbage then).  Call where the host synchronises anyway."""
    for key, ws in _pers_ws.items():
        info = (C.c_int * 4)()
        with torch.cuda.device(key):
            rc = lib().dvae_lstm_pers_check(ptr(ws), info, stream())
        if rc != 0:
            raise _lib.DvaeHipError(f"persistent LSTM launch gave up a bounded wait: pass {info[0]} (1 fwd, 2 bwd), "
                                    f"workgroup {info[1]}, step {info[2]}, wave {info[3]} (rc={rc})")


def _lstm_derived(derived, params, mode):
    """Per direction: (bias sum, W_ih^T, W_hh^T, fragment packs) — from the model's DerivedWeights when given, else
    computed on the spot (stand-alone use)."""
    if derived is not None:
        return list(derived)
    from .derived import lstm_local, lstm_local_pair
    if len(params) == 2 and params[0][1].shape[1] == 64:
        return lstm_local_pair(params, int(mode))
    return [lstm_local(wi, wh, bi, bh, int(mode)) for (wi, wh, bi, bh) in params]


class LstmLayerFn(torch.autograd.Function):
    """One nn.LSTM layer (1 or 2 directions) over frame-major rows: x[T*N, In] -> h[T*N, ndir*H].
    Reference: enc_lstm :163/:208, dec_lstm1 :172/:238, dec_lstm2 :193/:246.
    `derived`: list of derived.LstmDerived, one per direction (or None)."""

    @staticmethod
    def forward(ctx, x, T, N, w_ih, w_hh, b_ih, b_hh, w_ih_r, w_hh_r, b_ih_r, b_hh_r, derived=None, x16=None, emit16=False):
        """emit16 (bf16 compute mode, H a multiple of 512): returns (h, h16) — h16 the state as the frame kernels WROTE it
        (bf16), h its fp32 placeholder for autograd (see ConvBnActFn); otherwise h is a real fp32 tensor and h16 None."""
        x = x if x16 is None else x16                     # bf16 mode: the DATA of a placeholder input
        _ok(x, w_ih, w_hh, b_ih, b_hh, w_ih_r, w_hh_r, b_ih_r, b_hh_r, act16=(x,))
        L = lib()
        st = stream()
        dev = x.device
        R, In = x.shape
        H = w_hh.shape[1]
        ndir = 2 if w_ih_r is not None else 1
        ldh = ndir * H
        mode0 = current_mode()
        s16 = mode0 == MODE_BF16 and H % 512 == 0         # bf16 state / gate-gradient storage (frame kernels with S16)
        h_out = torch.empty((R, ldh), device=dev, dtype=torch.bfloat16 if s16 else torch.float32)
        esz = h_out.element_size()
        params = [(w_ih, w_hh, b_ih, b_hh), (w_ih_r, w_hh_r, b_ih_r, b_hh_r)][:ndir]
        dirs = (_lib.LstmDir * ndir)()
        # bf16 / fp32x3 compute modes: the recurrence runs on bf16 fragments (one plane / three planes) where such frame
        # kernels exist (H = 512, 1024); the H = 64 encoder recurrence (3 % of the FLOPs, weights resident in registers)
        # stays on the fp32 MFMA
        mode = current_mode()
        from .derived import lstm_pack_modes
        bf, bfb = lstm_pack_modes(mode, H)      # (forward, backward) precision of the recurrent product
        der = _lstm_derived(derived, params, mode)
        gates, cells = [], []
        # the two directions of a narrow layer (H = 64): ONE input projection with N = 8H into gates kept side by side
        cat = der[0].cat if (ndir == 2 and der[0].cat is not None and der[1].cat is der[0].cat) else None
        ldg = 8 * H if cat is not None else 4 * H
        if cat is not None:
            gall = torch.empty((R, ldg), device=dev, dtype=torch.float32)
            gemm(x, cat.w_ih, gall, cat.bias, R, ldg, In, In, In, ldg, True, True, mode=mode)
        for d, (wi, wh, bi, bh) in enumerate(params):
            if cat is not None:
                g = gall[:, d * 4 * H:(d + 1) * 4 * H]
            else:
                g = torch.empty((R, 4 * H), device=dev, dtype=torch.float32)
                gemm(x, wi if der[d].w_ih16 is None else der[d].w_ih16, g, der[d].bias, R, 4 * H, In, In, In, 4 * H, True,
                     True, mode=mode)
            c = torch.empty((R, H), device=dev, dtype=torch.float32)
            gates.append(g)
            cells.append(c)
            dirs[d].gate_ld = ldg
            dirs[d].gates = ptr(g)
            dirs[d].w_hh = ptr(wh)
            dirs[d].w_packed = ptr(der[d].pack_f)
            dirs[d].packed_mode = bf
            dirs[d].h_out = h_out.data_ptr() + esz * d * H
            dirs[d].c_all = ptr(c)
            dirs[d].reverse = d
            dirs[d].state_bf16 = int(s16)
        pers = lstm_persistent_usable(N, H, bf, ndir)
        if pers:
            _pers_fill(dirs[0], dev)
        check(L.dvae_lstm_seq_fwd(dirs, ndir, T, N, H, ldh, st), "dvae_lstm_seq_fwd")
        ctx.save_for_backward(x, h_out, *gates, *cells, *[p for ps in params for p in ps])
        ctx.der = der
        ctx.cfg = (T, N, H, ndir, bfb, mode, s16)
        ctx.cat = cat
        if not s16:
            return (h_out, None) if emit16 else h_out
        if emit16:
            ctx.mark_non_differentiable(h_out)
            ctx.set_materialize_grads(False)
            return _placeholder(R, ldh, dev), h_out
        return h_out.float()          # stand-alone use in the bf16 mode: a real fp32 copy for the caller

    @staticmethod
    def backward(ctx, dh, *_):
        return LstmLayerFn._backward(ctx, dh)

    @staticmethod
    def _backward(ctx, dh):
        T, N, H, ndir, bf, mode, s16 = ctx.cfg
        sv = ctx.saved_tensors
        x, h_out = sv[0], sv[1]
        esz = 2 if s16 else 4                                 # bytes per element of h_out / dgates
        gates = sv[2:2 + ndir]
        cells = sv[2 + ndir:2 + 2 * ndir]
        flat = sv[2 + 2 * ndir:]
        params = [flat[4 * d:4 * d + 4] for d in range(ndir)]
        der = ctx.der
        L = lib()
        st = stream()
        dev = x.device
        R, In = x.shape
        ldh = ndir * H
        dh = dh.contiguous()
        dirs = (_lib.LstmDir * ndir)()
        keep = []
        dgs = []
        cat = getattr(ctx, "cat", None)
        ldg = 8 * H if cat is not None else 4 * H
        if cat is not None:        # (H = 64: fp32 gate gradients) both directions side by side, like the gates
            dgall = torch.empty((R, ldg), device=dev, dtype=torch.float32)
        for d in range(ndir):
            if cat is not None:
                dg = dgall[:, d * 4 * H:(d + 1) * 4 * H]
            else:
                dg = torch.empty((R, 4 * H), device=dev, dtype=torch.bfloat16 if s16 else torch.float32)
            dc = torch.empty((N, H), device=dev, dtype=torch.float32)
            keep.append(dc)
            dgs.append(dg)
            dirs[d].gate_ld = ldg
            dirs[d].gates = ptr(gates[d])
            dirs[d].w_hh = ptr(der[d].w_hh_t)       # [H, 4H]
            dirs[d].w_packed = ptr(der[d].pack_b)
            dirs[d].packed_mode = bf
            dirs[d].c_all = ptr(cells[d])
            dirs[d].dh_out = dh.data_ptr() + 4 * d * H
            dirs[d].dgates = ptr(dg)
            dirs[d].dc_ws = ptr(dc)
            dirs[d].reverse = d
            dirs[d].state_bf16 = int(s16)
        pers = lstm_persistent_usable(N, H, bf, ndir, bwd=True)
        pers_bias = pers
        if pers:
            _pers_fill(dirs[0], dev)
        if pers_bias:
            # the persistent launch also leaves the bias gradients (column sums of dG): no colsum pass below.  Every row
            # group STORES its share into its own slab (no atomics); the slabs are added to both bias gradients (nn.LSTM
            # keeps two) with the other k-split slabs, in a fixed order
            bi, bh = params[0][2], params[0][3]
            gbi, gbh = _grad_buf(bi), _grad_buf(bh)
            key = (gbi.data_ptr(), 4 * H, "pers_bias")
            store = _slab_store(_owner_of(bi))
            part = store.get(key)
            if part is None:
                part = store[key] = torch.zeros(_lib.PERS_BIAS_SLABS * 4 * H, device=dev, dtype=torch.float32)
            dirs[0].dbias_part = ptr(part)
        check(L.dvae_lstm_seq_bwd(dirs, ndir, T, N, H, ldh, st), "dvae_lstm_seq_bwd")
        if pers_bias:
            for gb, par in ((gbi, bi), (gbh, bh)):
                _defer_fold(gb, _owner_of(par), part.data_ptr(), 4 * H, _lib.PERS_BIAS_SLABS, n=4 * H)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty((R, In), device=dev, dtype=torch.float32)
            if cat is not None:
                # both directions' data gradients are ONE contraction over K = 8H
                gemm(dgall, cat.w_ih_t, dx, None, R, In, ldg, ldg, ldg, In, True, True, ACT_NONE, EPI_STORE, mode=mode)
            # a narrow input (dec_lstm1: In = 128, so R / 128 output tiles fill half the chip) is cut along K = 4H into slabs
            sk = _split_k(_tiles(R, In), 4 * H) if (ndir == 1 and cat is None and (R * In) % 4 == 0) else 1
            if sk > 1:
                gemm_split(dgs[0], der[0].w_ih_t, dx, None, R, In, 4 * H, 4 * H, 4 * H, In, True, True, ACT_NONE, sk, mode)
            for d in range(ndir if (cat is None and sk == 1) else 0):
                # dx = dgates @ W_ih contracts over 4H: against W_ih^T both operands are k-contiguous
                gemm(dgs[d], der[d].w_ih_t, dx, None, R, In, 4 * H, 4 * H, 4 * H, In, True, True, ACT_NONE,
                     EPI_STORE if d == 0 else EPI_ACCUM, mode=mode)
        with side_work(x, h_out, *dgs):
            if ndir == 2 and 4 * H < 1024:
                # the two directions of a narrow layer (H = 64 encoder BiLSTM): each weight gradient alone is a few output
                # tiles over 16 384 rows — the forward and the reverse direction's products share ONE launch
                Nout, K = params[0][0].shape
                sk = _split_k(2 * _tiles(Nout, K), R)
                wgrad_gemm_batched(dgs, [x, x], [params[0][0], params[1][0]], Nout, K, R, ldg, In, sk, mode)
                if T > 1:
                    rows = R - N
                    sk = _split_k(2 * _tiles(4 * H, H), rows)
                    wgrad_gemm_batched([dgs[0].data_ptr() + esz * N * ldg, dgs[1].data_ptr()],
                                       [h_out.data_ptr(), h_out.data_ptr() + esz * (N * ldh + H)],
                                       [params[0][1], params[1][1]], 4 * H, H, rows, ldg, ldh, sk, mode,
                                       flags=(A_BF16 | B_BF16) if s16 else 0)
                else:
                    _grad_buf(params[0][1]), _grad_buf(params[1][1])
                for d, (wi, wh, bi, bh) in enumerate(params):
                    if not pers_bias:
                        colsum_add(dgs[d], _grad_buf(bi), _grad_buf(bh), rows=R, cols=4 * H, ld=ldg)
            else:
              for d, (wi, wh, bi, bh) in enumerate(params):
                dg = dgs[d]
                linear_wgrad_acc(dg, x, _grad_buf(wi), mode=mode, owner=_owner_of(wi))
                gw = _grad_buf(wh)      # exists (zero) even when T == 1 leaves W_hh without a gradient
                if T > 1:
                    rows = R - N
                    # h_prev of frame t is h[t-1] (forward) / h[t+1] (reverse)
                    if d == 0:
                        a_ptr, b_ptr = dg.data_ptr() + esz * N * 4 * H, h_out.data_ptr()
                    else:
                        a_ptr, b_ptr = dg.data_ptr(), h_out.data_ptr() + esz * (N * ldh + H)
                    sk = _split_k(_tiles(4 * H, H), rows)
                    wgrad_gemm(a_ptr, b_ptr, gw, _owner_of(wh), 4 * H, H, rows, 4 * H, ldh, False, False, sk, mode,
                               flags=(A_BF16 | B_BF16) if s16 else 0)
                if not pers_bias:
                    colsum_add(dg, _grad_buf(bi), _grad_buf(bh))
        for (wi, wh, bi, bh) in params:
            _ready(wi, wh, bi, bh)
        del keep
        return (dx,) + (None,) * 13


class LstmStack2Fn(torch.autograd.Function):
    """Two STACKED unidirectional LSTM layers of equal hidden size sharing their frame launches (dec_lstm2,
    disentangled_vae.py:193/:246; H = 1024): the sequence is cut into chunks of `CHUNK` frames; while layer 1 runs
    chunk c+1, layer 2 runs chunk c in the SAME launches (dvae_lstm_seq_*_range with step_shift), after layer 1's
    chunk-c outputs went through layer 2's input projection.  Same arithmetic per frame as two LstmLayerFn calls —
    only the launch schedule changes: T + CHUNK launches of 2x the workgroups instead of 2T, two workgroups resident
    per CU so one layer's load latency hides under the other's MFMAs.  Backward mirrors it (layer 2 ahead).
    `derived`: [LstmDerived of layer 1, LstmDerived of layer 2] (or None)."""

    CHUNK = int(os.environ.get("DVAE_LSTM_CHUNK", "-1"))      # frames per chunk; -1: two chunks (T/2), 0: off
    # measured at T = 128 (ms/step): unstacked 34.35, chunks of 16 / 32 / 64 frames 34.25 / 34.06 / 34.03
    # (measured alternatives: the trailing layer on a second HIP stream inside the graph, 35.0 ms/step against 34.0
    # fused and 34.2 unstacked; an idle skew of the trailing layer's workgroups: slower by exactly the skew)

    @staticmethod
    def chunk(T):
        return T // 2 if LstmStack2Fn.CHUNK < 0 else LstmStack2Fn.CHUNK

    @staticmethod
    def usable(T, H, layers, bidirectional):
        Tc = LstmStack2Fn.chunk(T)
        return layers == 2 and not bidirectional and H % 512 == 0 and Tc > 0 and T % Tc == 0 and T // Tc >= 2

    @staticmethod
    def forward(ctx, x, T, N, w_ih1, w_hh1, b_ih1, b_hh1, w_ih2, w_hh2, b_ih2, b_hh2, derived=None, x16=None, emit16=False):
        """emit16: as LstmLayerFn.forward — (h2 placeholder, h2 in bf16) in the bf16 compute mode."""
        x = x if x16 is None else x16                     # bf16 mode: the DATA of a placeholder input
        _ok(x, w_ih1, w_hh1, b_ih1, b_hh1, w_ih2, w_hh2, b_ih2, b_hh2, act16=(x,))
        L, st, dev = lib(), stream(), x.device
        R, In = x.shape
        H = w_hh1.shape[1]
        Tc = LstmStack2Fn.chunk(T)
        mode = current_mode()
        from .derived import lstm_pack_modes
        bf, bfb = lstm_pack_modes(mode, H)
        der = _lstm_derived(derived, [(w_ih1, w_hh1, b_ih1, b_hh1), (w_ih2, w_hh2, b_ih2, b_hh2)], mode)
        f32 = dict(device=dev, dtype=torch.float32)
        g1, g2 = torch.empty((R, 4 * H), **f32), torch.empty((R, 4 * H), **f32)
        c1, c2 = torch.empty((R, H), **f32), torch.empty((R, H), **f32)
        s16 = mode == MODE_BF16                               # bf16 state / gate-gradient storage (usable(): H % 512 == 0)
        sdt = dict(device=dev, dtype=torch.bfloat16 if s16 else torch.float32)
        esz = 2 if s16 else 4
        h1, h2 = torch.empty((R, H), **sdt), torch.empty((R, H), **sdt)
        gemm(x, w_ih1 if der[0].w_ih16 is None else der[0].w_ih16, g1, der[0].bias, R, 4 * H, In, In, In, 4 * H, True, True,
             mode=mode)
        dirs = (_lib.LstmDir * 2)()
        for d, (g, wh, h, c) in enumerate(((g1, w_hh1, h1, c1), (g2, w_hh2, h2, c2))):
            dirs[d].gates, dirs[d].w_hh, dirs[d].w_packed = ptr(g), ptr(wh), ptr(der[d].pack_f)
            dirs[d].h_out, dirs[d].c_all = ptr(h), ptr(c)
            dirs[d].reverse, dirs[d].packed_mode, dirs[d].step_shift = 0, bf, (Tc if d == 1 else 0)
            dirs[d].state_bf16 = int(s16)
        rows = Tc * N
        if lstm_persistent_usable(N, H, bf):
            # the forward recurrences as two W_hh-resident launches (one per layer), layer 2's whole input projection in
            # between; the backward pass keeps the stacked per-frame launches below
            one = (_lib.LstmDir * 1)()
            for d, (g, wh, h, c) in enumerate(((g1, w_hh1, h1, c1), (g2, w_hh2, h2, c2))):
                if d == 1:
                    gemm(h1, w_ih2 if der[1].w_ih16 is None else der[1].w_ih16, g2, der[1].bias, R, 4 * H, H, H, H, 4 * H,
                         True, True, mode=mode)
                one[0].gates, one[0].w_hh, one[0].w_packed = ptr(g), ptr(wh), ptr(der[d].pack_f)
                one[0].h_out, one[0].c_all = ptr(h), ptr(c)
                one[0].reverse, one[0].packed_mode, one[0].step_shift, one[0].state_bf16 = 0, bf, 0, int(s16)
                _pers_fill(one[0], dev)
                check(L.dvae_lstm_seq_fwd(one, 1, T, N, H, H, st), "dvae_lstm_seq_fwd")
        else:
            for c0 in range(0, T + Tc, Tc):
                if c0 >= Tc:    # layer 1 finished frames [c0-Tc, c0): their rows go through layer 2's input projection
                    r0 = (c0 - Tc) * N
                    gemm(h1.data_ptr() + esz * r0 * H, w_ih2 if der[1].w_ih16 is None else der[1].w_ih16,
                         g2.data_ptr() + 4 * r0 * 4 * H, der[1].bias, rows, 4 * H, H, H, H, 4 * H, True, True, mode=mode,
                         flags=A_BF16 if s16 else 0)
                check(L.dvae_lstm_seq_fwd_range(dirs, 2, T, N, H, H, c0, c0 + Tc, st), "dvae_lstm_seq_fwd_range")
        ctx.save_for_backward(x, h1, h2, g1, g2, c1, c2, w_ih1, w_hh1, b_ih1, b_hh1, w_ih2, w_hh2, b_ih2, b_hh2)
        ctx.der = der
        ctx.cfg = (T, N, H, bfb, mode, s16)
        if not s16:
            return (h2, None) if emit16 else h2
        if emit16:
            ctx.mark_non_differentiable(h2)
            ctx.set_materialize_grads(False)
            return _placeholder(R, H, dev), h2
        return h2.float()

    @staticmethod
    def backward(ctx, dh2, *_):
        T, N, H, bf, mode, s16 = ctx.cfg
        esz = 2 if s16 else 4
        (x, h1, h2, g1, g2, c1, c2, w_ih1, w_hh1, b_ih1, b_hh1, w_ih2, w_hh2, b_ih2, b_hh2) = ctx.saved_tensors
        der = ctx.der
        L, st, dev = lib(), stream(), x.device
        R, In = x.shape
        Tc = LstmStack2Fn.chunk(T)
        f32 = dict(device=dev, dtype=torch.float32)
        dh2 = dh2.contiguous()
        sdt = dict(device=dev, dtype=torch.bfloat16 if s16 else torch.float32)
        dg1, dg2 = torch.empty((R, 4 * H), **sdt), torch.empty((R, 4 * H), **sdt)
        dh1 = torch.empty((R, H), **f32)                   # gradient w.r.t. layer 1's outputs = layer 2's dgrad
        dcs = [torch.empty((N, H), **f32) for _ in range(2)]
        # backward step s handles frame T-1-s.  Entry 0 = layer 2 (ahead), entry 1 = layer 1 (a chunk behind).
        dirs = (_lib.LstmDir * 2)()
        for d, (g, dd, c, dho, dg) in enumerate(((g2, der[1], c2, dh2, dg2), (g1, der[0], c1, dh1, dg1))):
            dirs[d].gates, dirs[d].w_hh, dirs[d].w_packed, dirs[d].c_all = ptr(g), ptr(dd.w_hh_t), ptr(dd.pack_b), ptr(c)
            dirs[d].dh_out, dirs[d].dgates, dirs[d].dc_ws = ptr(dho), ptr(dg), ptr(dcs[d])
            dirs[d].reverse, dirs[d].packed_mode, dirs[d].step_shift = 0, bf, (Tc if d == 1 else 0)
            dirs[d].state_bf16 = int(s16)
        rows = Tc * N
        for c0 in range(0, T + Tc, Tc):
            if c0 >= Tc:    # layer 2 finished backward steps [c0-Tc, c0) = frames [T-c0, T-c0+Tc): dgrad into dh1
                r0 = (T - c0) * N
                gemm(dg2.data_ptr() + esz * r0 * 4 * H, der[1].w_ih_t, dh1.data_ptr() + 4 * r0 * H, None,
                     rows, H, 4 * H, 4 * H, 4 * H, H, True, True, mode=mode, flags=A_BF16 if s16 else 0)
            check(L.dvae_lstm_seq_bwd_range(dirs, 2, T, N, H, H, c0, c0 + Tc, st), "dvae_lstm_seq_bwd_range")
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty((R, In), **f32)
            gemm(dg1, der[0].w_ih_t, dx, None, R, In, 4 * H, 4 * H, 4 * H, In, True, True, mode=mode)
        with side_work(x, h1, h2, dg1, dg2):
            for dg, inp, hh, wi, wh, bi, bh in ((dg2, h1, h2, w_ih2, w_hh2, b_ih2, b_hh2),
                                                (dg1, x, h1, w_ih1, w_hh1, b_ih1, b_hh1)):
                linear_wgrad_acc(dg, inp, _grad_buf(wi), mode=mode, owner=_owner_of(wi))
                gw = _grad_buf(wh)
                rws = R - N
                sk = _split_k(_tiles(4 * H, H), rws)
                wgrad_gemm(dg.data_ptr() + esz * N * 4 * H, hh, gw, _owner_of(wh), 4 * H, H, rws, 4 * H, H, False, False, sk,
                           mode, flags=(A_BF16 if s16 else 0) | (B_BF16 if hh.dtype == torch.bfloat16 else 0))
                colsum_add(dg, _grad_buf(bi), _grad_buf(bh))
        _ready(w_ih2, w_hh2, b_ih2, b_hh2)
        _ready(w_ih1, w_hh1, b_ih1, b_hh1)
        del dcs
        return (dx,) + (None,) * 13


# ----------------------------------------------------------------------------- layout
class FramesToMelFn(torch.autograd.Function):
    """[T*N, C] -> [N, C, T] (decode()'s final transpose, disentangled_vae.py:248)."""

    @staticmethod
    def forward(ctx, y, N, Cc, T):
        _ok(y)
        out = torch.empty((N, Cc, T), device=y.device, dtype=torch.float32)
        check(lib().dvae_frames_to_mel(ptr(y), ptr(out), N, Cc, T, stream()), "dvae_frames_to_mel")
        return out

    @staticmethod
    def backward(ctx, dout):
        return mel_to_frames(dout.contiguous()), None, None, None


class Permute102Fn(torch.autograd.Function):
    """x viewed as [A,B,C] -> [B,A,C], returned with `out_shape` (the reshape between frame-major LSTM rows
    and the flat [B, T*128] bottleneck, disentangled_vae.py:209,235)."""

    @staticmethod
    def forward(ctx, x, A, B, Cc, out_shape):
        _ok(x)
        ctx.dims = (A, B, Cc, tuple(x.shape))
        out = torch.empty(tuple(out_shape), device=x.device, dtype=torch.float32)
        check(lib().dvae_permute_102(ptr(x), ptr(out), A, B, Cc, stream()), "dvae_permute_102")
        return out

    @staticmethod
    def backward(ctx, dout):
        A, B, Cc, in_shape = ctx.dims
        dout = dout.contiguous()
        dx = torch.empty(in_shape, device=dout.device, dtype=torch.float32)
        check(lib().dvae_permute_102(ptr(dout), ptr(dx), B, A, Cc, stream()), "dvae_permute_102")
        return dx, None, None, None, None


# ----------------------------------------------------------------------------- latent / losses
class FanoutFn(torch.autograd.Function):
    """x -> n aliases of x, one per consumer: the gradients of the consumers are summed by ONE launch of dvae_sum_f32
    instead of n - 1 element-wise additions issued by the autograd engine."""

    @staticmethod
    def forward(ctx, x, n):
        ctx.set_materialize_grads(False)
        return tuple(x.view_as(x) for _ in range(n))

    @staticmethod
    def backward(ctx, *gs):
        gs = [g.contiguous() for g in gs if g is not None]
        if not gs:
            return None, None
        while len(gs) > 1:
            a, b = gs.pop(), gs.pop()
            c = gs.pop() if gs else None
            if a.numel() % 4 or a.dtype != torch.float32:
                gs.append(a + b if c is None else a + b + c)
                continue
            out = torch.empty_like(a)
            check(lib().dvae_sum_f32(ptr(a), ptr(b), ptr(c), ptr(out), a.numel(), stream()), "dvae_sum_f32")
            gs.append(out)
        return gs[0], None


def fanout(x, n):
    return FanoutFn.apply(x, n) if (x.requires_grad and torch.is_grad_enabled()) else (x,) * n


class LatentFn(torch.autograd.Function):
    """Style averaging (x2 head detached), three reparameterisations and the q_z concatenations
    (disentangled_vae.py:222-228, 252-272) in one kernel."""

    @staticmethod
    def forward(ctx, style, content, eps_c, eps_s, Bh, S, Cn):
        _ok(style, content, eps_c, eps_s)
        dev = style.device
        D = S + Cn
        z = torch.empty((2 * Bh, D), device=dev, dtype=torch.float32)
        q_mu = torch.empty_like(z)
        q_lv = torch.empty_like(z)
        s_mu = torch.empty((Bh, S), device=dev, dtype=torch.float32)
        s_lv = torch.empty_like(s_mu)
        check(lib().dvae_latent_fwd(ptr(style), ptr(content), ptr(eps_c), ptr(eps_s), ptr(z), ptr(q_mu), ptr(q_lv),
                                    ptr(s_mu), ptr(s_lv), Bh, S, Cn, stream()), "dvae_latent_fwd")
        ctx.save_for_backward(style, content, eps_c, eps_s)
        ctx.dims = (Bh, S, Cn)
        return z, q_mu, q_lv, s_mu, s_lv

    @staticmethod
    def backward(ctx, dz, dq_mu, dq_lv, ds_mu, ds_lv):
        style, content, eps_c, eps_s = ctx.saved_tensors
        Bh, S, Cn = ctx.dims
        c = lambda t: None if t is None else t.contiguous()
        dz, dq_mu, dq_lv, ds_mu, ds_lv = c(dz), c(dq_mu), c(dq_lv), c(ds_mu), c(ds_lv)
        dstyle = torch.empty_like(style)
        dcontent = torch.empty_like(content)
        check(lib().dvae_latent_bwd(ptr(style), ptr(content), ptr(eps_c), ptr(eps_s), ptr(dz), ptr(dq_mu),
                                    ptr(dq_lv), ptr(ds_mu), ptr(ds_lv), ptr(dstyle), ptr(dcontent), Bh, S, Cn,
                                    stream()), "dvae_latent_bwd")
        return dstyle, dcontent, None, None, None, None, None


class KlFn(torch.autograd.Function):
    """scale * sum(1 + lv - mu^2 - exp(lv))  (disentangled_vae.py:320-323)."""

    @staticmethod
    def forward(ctx, mu, lv, scale):
        mu, lv = mu.contiguous(), lv.contiguous()
        _ok(mu, lv)
        out = torch.empty((), device=mu.device, dtype=torch.float32)
        check(lib().dvae_kl_fwd(ptr(mu), ptr(lv), ptr(out), mu.numel(), scale, stream()), "dvae_kl_fwd")
        ctx.save_for_backward(mu, lv)
        ctx.scale = scale
        return out

    @staticmethod
    def backward(ctx, g):
        mu, lv = ctx.saved_tensors
        g = g.contiguous()
        dmu, dlv = torch.empty_like(mu), torch.empty_like(lv)
        check(lib().dvae_kl_bwd(ptr(mu), ptr(lv), ptr(g), ptr(dmu), ptr(dlv), mu.numel(), ctx.scale, stream()),
              "dvae_kl_bwd")
        return dmu, dlv, None


class L1SumFn(torch.autograd.Function):
    """scale * sum|x - y|  (F.l1_loss(reduction='sum').div(batch_size), disentangled_vae.py:314-318)."""

    @staticmethod
    def forward(ctx, x, y, scale):
        x, y = x.contiguous(), y.contiguous()
        _ok(x, y)
        L = lib()
        n = x.numel()
        out = torch.empty((), device=x.device, dtype=torch.float32)
        ws = torch.empty((L.dvae_l1_ws_bytes(n),), device=x.device, dtype=torch.uint8)
        check(L.dvae_l1_sum_fwd(ptr(x), ptr(y), ptr(out), ptr(ws), n, scale, stream()), "dvae_l1_sum_fwd")
        ctx.save_for_backward(x, y)
        ctx.scale = scale
        return out

    @staticmethod
    def backward(ctx, g):
        x, y = ctx.saved_tensors
        g = g.contiguous()
        dy = torch.empty_like(y)
        check(lib().dvae_l1_sum_bwd(ptr(x), ptr(y), ptr(g), ptr(dy), x.numel(), ctx.scale, stream()),
              "dvae_l1_sum_bwd")
        return None, dy, None


class LossGVAE2Fn(torch.autograd.Function):
    """The eight scalars of loss_functionGVAE2 (disentangled_vae.py:310-327) as ONE vector
    (LOSS, L1_x1, L1_x2, L1_x1hat, L1_x2hat, KL_z1, KL_z2, KL_style): two launches forward, one backward."""

    @staticmethod
    def _desc(x1, x2, r, q, scales):
        d = _lib.LossDesc()
        d.x1, d.x2 = ptr(x1), ptr(x2)
        d.recon1, d.recon2, d.recon1_hat, d.recon2_hat = (ptr(t) for t in r)
        d.q1_mu, d.q1_lv, d.q2_mu, d.q2_lv, d.s_mu, d.s_lv = (ptr(t) for t in q)
        d.n, d.nq, d.ns = x1.numel(), q[0].numel(), q[4].numel()
        d.l1_scale, d.kl_scale, d.style_scale, d.mse_cof, d.kl_cof = scales
        return d

    @staticmethod
    def forward(ctx, x1, x2, r1, r2, h1, h2, q1mu, q1lv, q2mu, q2lv, smu, slv, l1_scale, kl_scale, style_scale, mse_cof,
                kl_cof):
        ts = [t.contiguous() for t in (x1, x2, r1, r2, h1, h2, q1mu, q1lv, q2mu, q2lv, smu, slv)]
        _ok(*ts)
        for t in ts[1:6]:
            if t.numel() != ts[0].numel():
                raise ValueError("loss: inputs and reconstructions must have the same number of elements")
        if not (ts[6].numel() == ts[7].numel() == ts[8].numel() == ts[9].numel() and ts[10].numel() == ts[11].numel()):
            raise ValueError("loss: mu / logvar shapes do not match")
        scales = (float(l1_scale), float(kl_scale), float(style_scale), float(mse_cof), float(kl_cof))
        L = lib()
        out = torch.empty(8, device=ts[0].device, dtype=torch.float32)
        ws = torch.empty((L.dvae_loss_ws_bytes(ts[0].numel()),), device=ts[0].device, dtype=torch.uint8)
        d = LossGVAE2Fn._desc(ts[0], ts[1], ts[2:6], ts[6:12], scales)
        check(L.dvae_loss_fwd(C.byref(d), ptr(out), ptr(ws), stream()), "dvae_loss_fwd")
        ctx.save_for_backward(*ts)
        ctx.scales = scales
        return out

    @staticmethod
    def backward(ctx, g):
        ts = ctx.saved_tensors
        g = g.contiguous()
        need = ctx.needs_input_grad
        grads = [None] * 12
        for i in range(2, 12):
            if need[i]:
                grads[i] = torch.empty_like(ts[i])
        # mu / logvar gradients come in pairs
        for a, b in ((6, 7), (8, 9), (10, 11)):
            if (grads[a] is None) != (grads[b] is None):
                grads[a] = grads[a] if grads[a] is not None else torch.empty_like(ts[a])
                grads[b] = grads[b] if grads[b] is not None else torch.empty_like(ts[b])
        d = LossGVAE2Fn._desc(ts[0], ts[1], ts[2:6], ts[6:12], ctx.scales)
        check(lib().dvae_loss_bwd(C.byref(d), ptr(g), *[ptr(grads[i]) for i in range(2, 12)], stream()), "dvae_loss_bwd")
        return (None, None, *[grads[i] if need[i] else None for i in range(2, 12)], None, None, None, None, None)


class LossGVAE2FullFn(torch.autograd.Function):
    """LossGVAE2Fn on the UNSPLIT forward outputs (model.forward_full): rec / rec_hat [2*Bh, ...] and q_mu / q_lv
    [2*Bh, Cn] hold the x1 half first, so the kernels get the halves as pointer offsets and backward writes both halves of
    one full-size gradient tensor — autograd never sees a slice (whose backward is a zero-fill plus a copy)."""

    @staticmethod
    def _desc(x1, x2, rec, rec_hat, q_mu, q_lv, s_mu, s_lv, scales):
        d = _lib.LossDesc()
        n, nq = x1.numel(), q_mu.numel() // 2
        d.x1, d.x2 = ptr(x1), ptr(x2)
        d.recon1, d.recon2 = rec.data_ptr(), rec.data_ptr() + 4 * n
        d.recon1_hat, d.recon2_hat = rec_hat.data_ptr(), rec_hat.data_ptr() + 4 * n
        d.q1_mu, d.q2_mu = q_mu.data_ptr(), q_mu.data_ptr() + 4 * nq
        d.q1_lv, d.q2_lv = q_lv.data_ptr(), q_lv.data_ptr() + 4 * nq
        d.s_mu, d.s_lv = ptr(s_mu), ptr(s_lv)
        d.n, d.nq, d.ns = n, nq, s_mu.numel()
        d.l1_scale, d.kl_scale, d.style_scale, d.mse_cof, d.kl_cof = scales
        return d

    @staticmethod
    def forward(ctx, x1, x2, rec, rec_hat, q_mu, q_lv, s_mu, s_lv, l1_scale, kl_scale, style_scale, mse_cof, kl_cof):
        ts = [t.contiguous() for t in (x1, x2, rec, rec_hat, q_mu, q_lv, s_mu, s_lv)]
        _ok(*ts)
        if not (ts[0].numel() == ts[1].numel() and ts[2].numel() == ts[3].numel() == 2 * ts[0].numel()):
            raise ValueError("loss: reconstructions must hold the x1 and the x2 half")
        if not (ts[4].numel() == ts[5].numel() and ts[4].numel() % 2 == 0 and ts[6].numel() == ts[7].numel()):
            raise ValueError("loss: mu / logvar shapes do not match")
        scales = (float(l1_scale), float(kl_scale), float(style_scale), float(mse_cof), float(kl_cof))
        L = lib()
        out = torch.empty(8, device=ts[0].device, dtype=torch.float32)
        ws = torch.empty((L.dvae_loss_ws_bytes(ts[0].numel()),), device=ts[0].device, dtype=torch.uint8)
        d = LossGVAE2FullFn._desc(*ts, scales)
        check(L.dvae_loss_fwd(C.byref(d), ptr(out), ptr(ws), stream()), "dvae_loss_fwd")
        ctx.save_for_backward(*ts)
        ctx.scales = scales
        return out

    @staticmethod
    def backward(ctx, g):
        ts = ctx.saved_tensors
        g = g.contiguous()
        grads = [torch.empty_like(t) for t in ts[2:]]            # rec, rec_hat, q_mu, q_lv, s_mu, s_lv
        d = LossGVAE2FullFn._desc(*ts, ctx.scales)
        n, nq = ts[0].numel(), ts[4].numel() // 2
        gp = lambda t, off=0: t.data_ptr() + 4 * off
        check(lib().dvae_loss_bwd(C.byref(d), ptr(g), gp(grads[0]), gp(grads[0], n), gp(grads[1]), gp(grads[1], n),
                                  gp(grads[2]), gp(grads[3]), gp(grads[2], nq), gp(grads[3], nq), gp(grads[4]), gp(grads[5]),
                                  stream()), "dvae_loss_bwd")
        need = ctx.needs_input_grad
        return (None, None, *[grads[i] if need[i + 2] else None for i in range(6)], None, None, None, None, None)


def prof_enable(family: int):
    check(lib().dvae_prof_enable(family), "dvae_prof_enable")


def prof_collect_tags(max_tags: int = 64):
    """Per-instantiation breakdown [(tag dict, ms, launches, flops)] of the profiled family; call before prof_collect()."""
    tags, ms = (C.c_uint * max_tags)(), (C.c_double * max_tags)()
    n_l, fl, by = (C.c_int64 * max_tags)(), (C.c_double * max_tags)(), (C.c_double * max_tags)()
    n = lib().dvae_prof_collect_tags(tags, ms, n_l, fl, by, max_tags)
    if n < 0:
        raise _lib.DvaeHipError(f"dvae_prof_collect_tags failed: rc={n}")
    out = []
    for i in range(n):
        t = int(tags[i])
        bns, a16, b16 = (t >> 19) & 1, (t >> 17) & 1, (t >> 18) & 1
        if (t >> 10) & 7 == 3:      # the 256 x 256 LDS-DMA tile of the bf16 mode (gemm256.hip)
            name = f"gemm_bf16_256_kernel<A_KC={t & 1}, B_KC={(t >> 1) & 1}, BNS={bns}> tap_mode={(t >> 15) & 3}"
        elif (t >> 10) & 7 == 1:    # the 256 x 128 tile, one workgroup per CU
            kn = "gemm_bf16_tall_kernel" if (t >> 13) & 3 == 1 else "gemm_x3_tall_kernel"
            name = f"{kn}<A_KC={t & 1}, B_KC={(t >> 1) & 1}, BNS={bns}> tap_mode={(t >> 15) & 3}"
        else:
            name = (f"gemm_f32_kernel<A_KC={t & 1}, B_KC={(t >> 1) & 1}, NTW={(t >> 2) & 3}, BK={(t >> 4) & 63}, "
                    f"WG={(t >> 10) & 7}, MODE={(t >> 13) & 3}, BNS={bns}, A16={a16}, B16M={b16}> tap_mode={(t >> 15) & 3}")
        out.append({"tag": t, "kernel": name, "ms": float(ms[i]), "launches": int(n_l[i]), "flops": float(fl[i]),
                    "bytes": float(by[i])})
    return sorted(out, key=lambda d: -d["ms"])


def prof_collect():
    ms, n, fl = C.c_double(), C.c_int64(), C.c_double()
    check(lib().dvae_prof_collect(C.byref(ms), C.byref(n), C.byref(fl)), "dvae_prof_collect")
    return ms.value, n.value, fl.value


if os.environ.get("DVAE_DETERMINISTIC", "0") == "1":
    set_deterministic(True)
