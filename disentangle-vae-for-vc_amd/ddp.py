"""Data parallelism for the training step: one process per GPU, minibatch sharded across ranks,
ONE exchange per step — a bucketed sum all-reduce of the flat gradient buffer over RCCL/xGMI
(backend "nccl" is RCCL on ROCm), overlapped with the rest of backward.

The reference is single-device (no DDP/NCCL anywhere); this is new functionality
designed for MI355X:
  * gradients already live in one flat buffer laid out in backward-completion order
    (DisentangledVAE.backward_param_order), so a bucket is a contiguous slice: no packing copies;
  * xGMI is point-to-point (7 links x ~153 GB/s per GPU), ring collectives are per-link bound, so
    buckets are few and large (default 64 MiB: ~6 buckets for the 380 MB of fp32 gradients at T=128);
  * a bucket's all-reduce is enqueued as soon as the last parameter in it has its gradient
    (ops.grad_ready_hook), on RCCL's own stream, and joins the compute stream only before Adam;
  * BatchNorm statistics stay rank-local and the loss stays divided by the LOCAL batch size; the
    summed gradients are scaled by 1/world inside the Adam kernel — identical to a global
    `sum / B_global`.
Works with any torch.distributed backend (the CPU tests use gloo on CPU tensors).
"""
from __future__ import annotations

from typing import Dict, List, Sequence, Tuple

import torch
import torch.distributed as dist

from . import ops


class GradReducer:
    """mode "all_reduce" (default): every bucket is summed over the ranks in place (dist.all_reduce) and every rank
    runs the full Adam launch over all parameters.
    mode "rs_ag" (direct reduce-scatter + all-gather with a sharded optimizer): every bucket is
    REDUCE-SCATTERED — rank r receives the sum of the r-th 1/world of the bucket — `step()` runs Adam on the slices this rank
    owns (1/world of the optimizer traffic per rank) and ALL-GATHERS the updated parameters.  Same bytes on the wire as
    a ring all-reduce (which is a reduce-scatter followed by an all-gather), but the all-gather moves PARAMETERS after
    the update, so the 7 x 4 bytes per parameter of Adam are paid once per node, not once per GPU.  Moments of the
    slices a rank does not own go stale on that rank: `gather_moments()` (before a checkpoint) collects them.
    issue "hook" (default): a bucket's collective is enqueued from the autograd hook as soon as its last gradient is
    complete (overlap with the rest of backward).  issue "finish": all collectives are enqueued after backward — no
    overlap, and no RCCL kernel resident beside the W_hh-resident recurrences of backward (which want every CU)."""

    def __init__(self, flat_grad: torch.Tensor, names: Sequence[str], params: Sequence[torch.Tensor],
                 offsets: Dict[str, int], bucket_bytes: int = 64 << 20, group=None, mode: str = "all_reduce",
                 issue: str = "hook"):
        if mode not in ("all_reduce", "rs_ag") or issue not in ("hook", "finish"):
            raise ValueError("GradReducer: mode is 'all_reduce' or 'rs_ag', issue is 'hook' or 'finish'")
        self.flat = flat_grad
        self.params = list(params)
        self.group = group
        self.mode, self.issue = mode, issue
        self.world_size = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        # bucket boundaries fall on multiples of `q` elements: 16-byte alignment, and in rs_ag mode equal, aligned shards
        q = 4 * self.world_size if mode == "rs_ag" else 4
        total = flat_grad.numel()
        if mode == "rs_ag" and total % q:
            raise ValueError(f"rs_ag: the flat buffer ({total} elements) must be a multiple of 4 x world = {q} "
                             "(optim.FlatAdam pads to 32: world sizes 1, 2, 4, 8)")
        cap = max(q, bucket_bytes // 4)
        # contiguous buckets in buffer order (= backward completion order)
        self.buckets: List[Tuple[int, int]] = []
        ordered = sorted(zip(names, params), key=lambda np_: offsets[np_[0]])
        spans = [(offsets[n], offsets[n] + (p.numel() + 3) // 4 * 4, p) for n, p in ordered]
        lo = 0
        for i, (a, e, p) in enumerate(spans):
            last = i == len(spans) - 1
            end = total if last else min(total, (e + q - 1) // q * q)
            if end - lo >= cap or last:
                if end > lo:
                    self.buckets.append((lo, end))
                lo = end
        # a bucket is complete when every parameter that OVERLAPS it has its gradient (a parameter may straddle a rounded
        # boundary and then belongs to two buckets)
        self.buckets_of: Dict[int, List[int]] = {}
        self.pending_init: List[int] = [0] * len(self.buckets)
        for a, e, p in spans:
            bs = [b for b, (blo, bhi) in enumerate(self.buckets) if a < bhi and e > blo]
            self.buckets_of[id(p)] = bs
            for b in bs:
                self.pending_init[b] += 1
        self.bucket_of = {k: v[-1] for k, v in self.buckets_of.items() if v}      # (kept: the bucket a parameter ENDS in)
        self.pending = list(self.pending_init)
        self.next_bucket = 0    # buckets are LAUNCHED in buffer order on every rank, whatever order they complete in
        self.works = []
        self.active = False
        self.force = False      # issue the collective even with one rank (tests of the RCCL path)
        # a backend without device-memory collectives (gloo on GPU tensors: the shared-GPU functional checks) gets the
        # buckets staged through host memory in finish(); RCCL ("nccl") reduces the device slices in place
        self.backend = dist.get_backend(group) if dist.is_initialized() else None
        self.staged = bool(flat_grad.is_cuda and self.backend is not None and self.backend != "nccl")
        self._deferred: List[int] = []
        # buckets launched from the autograd hook (overlapped with backward) / left over for finish(), cumulative
        self.stats = {"hook": 0, "finish": 0, "steps": 0}

    # ---- shards (rs_ag)
    def shard(self, b: int, rank: int = None) -> Tuple[int, int]:
        """[lo, hi) of bucket b that `rank` (default: this rank) owns in rs_ag mode."""
        lo, hi = self.buckets[b]
        per = (hi - lo) // self.world_size
        r = self.rank if rank is None else rank
        return lo + r * per, lo + (r + 1) * per

    def begin(self):
        self.pending = list(self.pending_init)
        self.next_bucket = 0
        self.works = []
        self.active = True
        ops.grad_ready_hook = self._ready
        # the k-split slabs of a weight gradient (ops.wgrad_gemm) must be summed into the flat buffer before its bucket leaves:
        # per reported parameter when buckets go out from the hooks, once for everything in finish() otherwise
        ops.fold_on_ready = self.issue == "hook"

    def _collective_on(self) -> bool:
        return self.world_size > 1 or (dist.is_initialized() and self.force)

    def _launch(self, b):
        lo, hi = self.buckets[b]
        if self.staged:
            self._deferred.append(b)
            return
        if not self._collective_on():
            return
        if self.mode == "rs_ag" and self.backend == "nccl":
            slo, shi = self.shard(b)
            # in place: the output is this rank's slice of the input
            self.works.append(dist.reduce_scatter_tensor(self.flat[slo:shi], self.flat[lo:hi], op=dist.ReduceOp.SUM,
                                                         group=self.group, async_op=True))
        else:
            # (a backend without reduce-scatter — gloo — sums the whole bucket: the owned slice holds the same values)
            self.works.append(dist.all_reduce(self.flat[lo:hi], op=dist.ReduceOp.SUM, group=self.group,
                                              async_op=True))

    def _ready(self, p):
        if not self.active:
            return
        for b in self.buckets_of.get(id(p), ()):
            self.pending[b] -= 1
        if self.issue != "hook":
            return
        # Collectives pair up across ranks by ISSUE order, so bucket b goes out only after buckets 0..b-1 have: a rank
        # whose hooks fire in another order (or that skips a parameter) still issues the same sequence as its peers.
        # The flat buffer is laid out in backward-completion order, so in the normal case nothing waits.
        while self.next_bucket < len(self.buckets) and self.pending[self.next_bucket] == 0:
            self.stats["hook"] += 1
            self._launch(self.next_bucket)
            self.next_bucket += 1

    def finish(self):
        """Enqueue whatever was not triggered (parameters without gradient this step) and join."""
        ops.grad_ready_hook = None
        ops.fold_on_ready = True
        ops.join_side()
        for ow in {id(getattr(p, "_dvae_owner", None)): getattr(p, "_dvae_owner", None) for p in self.params}.values():
            if ow is not None:
                ops.fold_pending(ow)      # whatever was not summed from a hook: in front of the collectives below
        self.active = False
        self.stats["steps"] += 1
        for b in range(self.next_bucket, len(self.buckets)):
            self.stats["finish"] += 1
            self._launch(b)
        self.next_bucket = len(self.buckets)
        for w in self.works:
            w.wait()
        self.works = []
        if self._deferred:
            torch.cuda.current_stream().synchronize()
            for b in self._deferred:
                lo, hi = self.buckets[b]
                host = self.flat[lo:hi].cpu()
                dist.all_reduce(host, op=dist.ReduceOp.SUM, group=self.group)
                self.flat[lo:hi].copy_(host)
            self._deferred = []

    # ---- sharded optimizer step (rs_ag)
    def step(self, optimizer):
        """After finish(): Adam on the slices this rank owns — `optimizer.step_range(lo, hi, grad_scale, tick)` once per
        bucket — then the all-gather of the updated parameters (`optimizer.flat_p`), bucket by bucket, each enqueued as
        soon as its slice is updated.  In "all_reduce" mode this is the plain full step."""
        scale = 1.0 / self.world_size
        if self.mode != "rs_ag":
            optimizer.step(grad_scale=scale)
            return
        flat_p = optimizer.flat_p
        works = []
        for b, (lo, hi) in enumerate(self.buckets):
            slo, shi = self.shard(b)
            optimizer.step_range(slo, shi, scale, tick=(b == 0))
            if not self._collective_on():
                continue
            if flat_p.is_cuda and self.backend != "nccl":
                torch.cuda.current_stream().synchronize()
                host = flat_p[slo:shi].cpu()
                parts = [torch.empty_like(host) for _ in range(self.world_size)]
                dist.all_gather(parts, host, group=self.group)
                flat_p[lo:hi].copy_(torch.cat(parts))
            elif self.backend == "nccl":
                works.append(dist.all_gather_into_tensor(flat_p[lo:hi], flat_p[slo:shi], group=self.group, async_op=True))
            else:
                parts = [flat_p[self.shard(b, r)[0]:self.shard(b, r)[1]] for r in range(self.world_size)]
                tmp = [torch.empty_like(x) for x in parts]
                dist.all_gather(tmp, flat_p[slo:shi].clone(), group=self.group)
                for dst, src in zip(parts, tmp):
                    dst.copy_(src)
        for w in works:
            w.wait()

    def gather_moments(self, optimizer):
        """rs_ag: every rank holds current Adam moments only for its own slices; collect all of them on every rank (call
        before optimizer.state_dict(), i.e. before a checkpoint)."""
        if self.mode != "rs_ag" or not self._collective_on():
            return
        for buf in (optimizer.exp_avg, optimizer.exp_avg_sq):
            for b, (lo, hi) in enumerate(self.buckets):
                slo, shi = self.shard(b)
                if self.backend == "nccl":
                    dist.all_gather_into_tensor(buf[lo:hi], buf[slo:shi], group=self.group)
                else:
                    src = buf[slo:shi].detach().cpu() if buf.is_cuda else buf[slo:shi].clone()
                    parts = [torch.empty_like(src) for _ in range(self.world_size)]
                    dist.all_gather(parts, src, group=self.group)
                    buf[lo:hi].copy_(torch.cat(parts))


def broadcast_parameters(flat_params: torch.Tensor, buffers: Sequence[torch.Tensor] = (), src: int = 0, group=None):
    """Rank `src`'s weights (and BatchNorm buffers) to every rank, once, before training."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    staged = flat_params.is_cuda and dist.get_backend(group) != "nccl"
    for t in [flat_params, *buffers]:
        if staged:
            host = t.detach().cpu()
            dist.broadcast(host, src=src, group=group)
            t.data.copy_(host)
        else:
            dist.broadcast(t, src=src, group=group)


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous shard [lo, hi) of a global batch for `rank` (global batch must divide evenly)."""
    if n_items % world:
        raise ValueError(f"global batch {n_items} is not divisible by world size {world}")
    per = n_items // world
    return rank * per, (rank + 1) * per
