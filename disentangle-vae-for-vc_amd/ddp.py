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
    def __init__(self, flat_grad: torch.Tensor, names: Sequence[str], params: Sequence[torch.Tensor],
                 offsets: Dict[str, int], bucket_bytes: int = 64 << 20, group=None):
        self.flat = flat_grad
        self.group = group
        self.world_size = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        cap = max(1, bucket_bytes // 4)
        # contiguous buckets in buffer order (= backward completion order)
        self.buckets: List[Tuple[int, int]] = []
        self.bucket_of: Dict[int, int] = {}
        self.pending_init: List[int] = []
        lo, count = 0, 0
        ordered = sorted(zip(names, params), key=lambda np_: offsets[np_[0]])
        for i, (n, p) in enumerate(ordered):
            end = offsets[n] + (p.numel() + 3) // 4 * 4
            self.bucket_of[id(p)] = len(self.buckets)
            count += 1
            last = i == len(ordered) - 1
            if end - lo >= cap or last:
                self.buckets.append((lo, flat_grad.numel() if last else end))
                self.pending_init.append(count)
                lo, count = end, 0
        self.pending = list(self.pending_init)
        self.next_bucket = 0    # buckets are LAUNCHED in buffer order on every rank, whatever order they complete in
        self.works = []
        self.active = False
        self.force = False      # issue the collective even with one rank (tests of the RCCL path)
        # a backend without device-memory collectives (gloo on GPU tensors: the shared-GPU functional checks) gets the
        # buckets staged through host memory in finish(); RCCL ("nccl") reduces the device slices in place
        self.staged = bool(flat_grad.is_cuda and dist.is_initialized() and dist.get_backend(group) != "nccl")
        self._deferred: List[int] = []
        # buckets launched from the autograd hook (overlapped with backward) / left over for finish(), cumulative
        self.stats = {"hook": 0, "finish": 0, "steps": 0}

    def begin(self):
        self.pending = list(self.pending_init)
        self.next_bucket = 0
        self.works = []
        self.active = True
        ops.grad_ready_hook = self._ready

    def _launch(self, b):
        lo, hi = self.buckets[b]
        if self.staged:
            self._deferred.append(b)
            return
        if self.world_size > 1 or (dist.is_initialized() and self.force):
            self.works.append(dist.all_reduce(self.flat[lo:hi], op=dist.ReduceOp.SUM, group=self.group,
                                              async_op=True))

    def _ready(self, p):
        if not self.active:
            return
        b = self.bucket_of.get(id(p))
        if b is None:
            return
        self.pending[b] -= 1
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
        ops.join_side()
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
