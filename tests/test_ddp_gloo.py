"""Data-parallel path on CPU: world_size 2, gloo backend, 127.0.0.1.

Checks (a) the bucketed reducer over a flat gradient buffer reproduces the SUM of per-rank gradients whatever
order parameters become ready in, and (b) the N-rank recipe used by bench.py / ddp.py — rank-local BatchNorm
statistics, loss divided by the local batch, summed gradients scaled by 1/world — equals the single-process
chunked step of the oracle.  The oracle supplies forward/backward here (tests may use it); the
reducer, the flat packing and the sharding are the product code under test."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _init(rank, world, port):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)


def _worker_reducer(rank, world, port, q):
    _init(rank, world, port)
    import dvae_amd  # noqa: F401
    from dvae_amd import ddp, ops
    from dvae_amd.optim import FlatAdam
    torch.manual_seed(0)
    ps = [(f"p{i}", torch.nn.Parameter(torch.zeros(n))) for i, n in enumerate((1000, 37, 4096, 5, 2048, 300))]
    opt = FlatAdam(ps, lr=1e-3)
    red = ddp.GradReducer(opt.flat_g, opt.names, opt.params, opt.offsets, bucket_bytes=8192)
    assert len(red.buckets) >= 3 and red.buckets[0][0] == 0 and red.buckets[-1][1] == opt.numel
    for a, b in zip(red.buckets[:-1], red.buckets[1:]):
        assert a[1] == b[0]
    for trial in range(2):
        opt.zero_grad()
        red.begin()
        order = list(range(len(ps))) if trial == 0 else [3, 0, 5, 1]      # trial 1: two params never report
        for i in range(len(ps)):
            ps[i][1].grad.add_(float(rank + 1) * (i + 1))
        for i in order:
            ops.grad_ready_hook(ps[i][1])
        red.finish()
        assert ops.grad_ready_hook is None
        for i, (_, p) in enumerate(ps):
            want = sum(float(r + 1) * (i + 1) for r in range(world))
            assert torch.allclose(p.grad, torch.full_like(p.grad, want)), (trial, i)
    ddp.broadcast_parameters(opt.flat_p)
    dist.barrier()
    if rank == 0:
        q.put("ok")
    dist.destroy_process_group()


def _worker_uneven_order(rank, world, port, q):
    """World 4: every rank's hooks fire in a DIFFERENT order and some ranks never report some parameters.  Buckets are
    unequal in size, so a rank that issued its collectives in completion order would pair a bucket with a peer's other
    bucket (size mismatch: error or hang) — the reducer must issue them in buffer order everywhere."""
    _init(rank, world, port)
    import random
    import dvae_amd  # noqa: F401
    from dvae_amd import ddp, ops
    from dvae_amd.optim import FlatAdam
    sizes = (900, 40, 3000, 8, 2500, 310, 1200, 64, 2048)
    ps = [(f"p{i}", torch.nn.Parameter(torch.zeros(n))) for i, n in enumerate(sizes)]
    opt = FlatAdam(ps, lr=1e-3)
    red = ddp.GradReducer(opt.flat_g, opt.names, opt.params, opt.offsets, bucket_bytes=6000)
    assert len(red.buckets) >= 4 and len({hi - lo for lo, hi in red.buckets}) > 1
    for trial in range(4):
        rng = random.Random(1000 * trial + rank)          # a different order per rank and trial
        order = list(range(len(ps)))
        rng.shuffle(order)
        if trial >= 2:
            order = order[: len(order) - 1 - rank % 3]    # ... and some parameters never report on some ranks
        opt.zero_grad()
        red.begin()
        for i in range(len(ps)):
            ps[i][1].grad.add_(float(rank + 1) * (i + 1) + trial)
        launched_before = red.stats["hook"]
        for i in order:
            ops.grad_ready_hook(ps[i][1])
        red.finish()
        assert red.next_bucket == len(red.buckets)
        assert red.stats["hook"] - launched_before <= len(red.buckets)
        for i, (_, p) in enumerate(ps):
            want = sum(float(r + 1) * (i + 1) + trial for r in range(world))
            assert torch.allclose(p.grad, torch.full_like(p.grad, want)), (trial, i, rank)
    dist.barrier()
    if rank == 0:
        q.put("ok")
    dist.destroy_process_group()


def _worker_step_equivalence(rank, world, port, q):
    _init(rank, world, port)
    import dvae_amd  # noqa: F401
    from dvae_amd import ddp
    from dvae_amd.optim import FlatAdam
    from oracle.dvae_ref import RefTrainer, chunked_step_grads, loss_gvae2
    from oracle.fill import fill_state_dict, synthetic_eps, synthetic_pair
    Bg, T = 4, 64
    per = Bg // world
    x1, x2 = synthetic_pair(Bg, T, 21)
    eps = synthetic_eps(Bg, seed=22)
    tr = RefTrainer(per, n_frames=T)
    tr.model.load_state_dict(fill_state_dict(tr.model.state_dict()))
    tr.model.train()
    opt = FlatAdam(list(tr.model.named_parameters()), lr=1e-4)
    red = ddp.GradReducer(opt.flat_g, opt.names, opt.params, opt.offsets, bucket_bytes=32 << 20)
    lo, hi = ddp.shard_range(Bg, rank, world)
    opt.zero_grad()
    red.begin()
    outs = tr.model(x1[lo:hi], x2[lo:hi], tuple(e[lo:hi] for e in eps))
    loss_gvae2(x1[lo:hi], x2[lo:hi], outs, per)[0].backward()      # rank-local BN stats, loss / local batch
    red.finish()                                                    # SUM over ranks
    assert opt.views_intact()
    got = {n: p.grad.detach().clone() / world for n, p in tr.model.named_parameters()}   # Adam's grad_scale
    if rank == 0:
        ref_tr = RefTrainer(per, n_frames=T)
        ref_tr.model.load_state_dict(fill_state_dict(ref_tr.model.state_dict()))
        ref_tr.model.train()
        want = chunked_step_grads(ref_tr, x1, x2, eps, world)
        worst = 0.0
        for n in want:
            denom = float(want[n].norm()) + 1e-6
            worst = max(worst, float((got[n] - want[n]).norm()) / denom if denom > 1e-3 else 0.0)
        q.put(worst)
    dist.barrier()
    dist.destroy_process_group()


class _CpuAdam:
    """TEST stand-in for FlatAdam's device launch on CPU tensors (the product's Adam runs only on the HIP device): the same
    flat buffers, `step` / `step_range` with torch arithmetic — what GradReducer.step drives."""

    def __init__(self, opt, lr=1e-2, b1=0.9, b2=0.999, eps=1e-8):
        self.o, self.lr, self.b1, self.b2, self.eps, self.t = opt, lr, b1, b2, eps, 0
        self.flat_p, self.exp_avg, self.exp_avg_sq = opt.flat_p, opt.exp_avg, opt.exp_avg_sq
        self.calls = []

    def step(self, grad_scale=1.0):
        self.step_range(0, self.o.numel, grad_scale, tick=True)

    def step_range(self, lo, hi, grad_scale=1.0, tick=True):
        if tick:
            self.t += 1
        self.calls.append((lo, hi, tick))
        g = self.o.flat_g[lo:hi] * grad_scale
        m, v, p = self.exp_avg[lo:hi], self.exp_avg_sq[lo:hi], self.flat_p[lo:hi]
        m.mul_(self.b1).add_(g, alpha=1 - self.b1)
        v.mul_(self.b2).addcmul_(g, g, value=1 - self.b2)
        bc1, bc2 = 1 - self.b1 ** self.t, 1 - self.b2 ** self.t
        p.sub_(self.lr / bc1 * m / (v.sqrt() / bc2 ** 0.5 + self.eps))


def _worker_rs_ag(rank, world, port, q):
    """World 4: three optimizer steps with the sharded path (reduce-scatter -> Adam on 1/world -> all-gather) against the
    all-reduce path on the same per-rank gradients: parameters equal to 1e-6 on every rank, each rank updated exactly its
    own slices, the gathered moments equal the all-reduce path's."""
    _init(rank, world, port)
    import dvae_amd  # noqa: F401
    from dvae_amd import ddp, ops
    from dvae_amd.optim import FlatAdam
    sizes = (900, 40, 3000, 8, 2500, 310, 1200, 64, 2048, 37)

    def build(mode, issue="hook"):
        torch.manual_seed(5)
        ps = [(f"p{i}", torch.nn.Parameter(torch.randn(n))) for i, n in enumerate(sizes)]
        opt = FlatAdam(ps, lr=1e-2)
        assert opt.numel % 32 == 0 and opt.numel >= opt.n_used
        red = ddp.GradReducer(opt.flat_g, opt.names, opt.params, opt.offsets, bucket_bytes=6000, mode=mode, issue=issue)
        return ps, opt, red, _CpuAdam(opt)

    A = build("all_reduce")
    S = build("rs_ag")
    F = build("rs_ag", issue="finish")
    assert len(S[2].buckets) >= 4
    for lo, hi in S[2].buckets:
        assert (hi - lo) % (4 * world) == 0                       # equal, 16-byte aligned shards
    assert S[2].buckets[0][0] == 0 and S[2].buckets[-1][1] == S[1].numel
    own = sum(S[2].shard(b)[1] - S[2].shard(b)[0] for b in range(len(S[2].buckets)))
    assert own * world == S[1].numel
    for step in range(3):
        g = torch.Generator().manual_seed(100 * step + rank)      # a different gradient per rank and step
        grads = [torch.randn(n, generator=g) for n in sizes]
        for ps, opt, red, adam in (A, S, F):
            opt.zero_grad()
            red.begin()
            order = list(range(len(ps)))
            import random
            random.Random(step * 10 + rank).shuffle(order)          # hooks fire in a different order on every rank
            for i in order:
                ps[i][1].grad.add_(grads[i])
                ops.grad_ready_hook(ps[i][1])
            red.finish()
            red.step(adam)
        assert F[2].stats["hook"] == 0                            # issue="finish": nothing left the hooks
        for name, X in (("rs_ag", S), ("rs_ag/finish", F)):
            d = float((A[1].flat_p - X[1].flat_p).abs().max())
            assert d <= 1e-6, (name, step, rank, d)
    # every rank ran Adam on its own slices only: one call per bucket, 1/world of the elements, one tick per step
    calls = S[3].calls
    assert len(calls) == 3 * len(S[2].buckets) and sum(c[2] for c in calls) == 3
    assert sum(c[1] - c[0] for c in calls) * world == 3 * S[1].numel
    S[2].gather_moments(S[1])
    assert float((A[1].exp_avg - S[1].exp_avg).abs().max()) <= 1e-6
    assert float((A[1].exp_avg_sq - S[1].exp_avg_sq).abs().max()) <= 1e-6
    dist.barrier()
    if rank == 0:
        q.put("ok")
    dist.destroy_process_group()


def _run(fn, world=2):
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=fn, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
    for p in procs:
        assert p.exitcode == 0, f"worker exit code {p.exitcode}"
    return q.get()


def test_bucketed_reducer_world2_gloo():
    assert _run(_worker_reducer) == "ok"


def test_reducer_world4_uneven_ready_order_gloo():
    assert _run(_worker_uneven_order, world=4) == "ok"


def test_two_rank_step_equals_chunked_oracle_step():
    worst = _run(_worker_step_equivalence)
    assert worst < 1e-4, worst


def test_sharded_optimizer_rs_ag_world4_gloo():
    assert _run(_worker_rs_ag, world=4) == "ok"
