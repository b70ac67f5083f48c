"""Rank process of tests/test_hip_ddp.py::test_two_rank_step_*: TWO ranks of the data-parallel step on the HIP kernels.
The GPU box has one GPU, so both ranks share cuda:0 and exchange through gloo (ddp.GradReducer stages the buckets through
host memory for a backend without device collectives; the persistent recurrences are off because two whole-chip grids
cannot both be resident).  Everything else is the product path: sharded global batch, rank-local BatchNorm, loss / local
batch, bucketed sum from the autograd hooks, 1/world inside Adam.

env: RANK, WORLD_SIZE, MASTER_ADDR, MASTER_PORT;  argv: <out_dir> <lr> <steps>"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    out_dir, lr, steps = sys.argv[1], float(sys.argv[2]), int(sys.argv[3])
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import dvae_amd
    from dvae_amd import ddp, ops
    from oracle.fill import fill_state_dict, synthetic_eps, synthetic_pair     # deterministic weights / inputs only
    ops.LSTM_PERSISTENT = False
    Bg, T = 4, 64
    per = Bg // world
    w = dvae_amd.ConvolutionalMulVAE("VCTK", T, 80, 32, lr, 0.01, 500, False, batch_size=per, speaker_size=4,
                                     device=dev, latent_dim=32, mse_cof=10, kl_cof=10)
    if rank == 0:          # only rank 0 holds the reference weights: the broadcast has to deliver them
        w.model.load_state_dict(fill_state_dict(w.model.state_dict()))
    w.model.train()
    opt = w.optimizer
    opt.fold_zero_grad = False           # this test reads the summed gradients AFTER the step
    ddp.broadcast_parameters(opt.flat_p, list(w.model.buffers()))
    red = ddp.GradReducer(opt.flat_g, opt.names, opt.params, opt.offsets, bucket_bytes=32 << 20)
    assert red.staged and red.world_size == world
    w.attach_reducer(red)
    lo, hi = ddp.shard_range(Bg, rank, world)
    losses = []
    for i in range(steps):
        x1, x2 = synthetic_pair(Bg, T, 21 + i)
        eps = synthetic_eps(Bg, seed=22 + i)
        w.model.eps_override = tuple(e[lo:hi] for e in eps)
        losses.append(list(w.step(x1[lo:hi].cuda(), x2[lo:hi].cuda(), None, train=True)))
    torch.cuda.synchronize()
    assert opt.views_intact()
    res = {"rank": rank, "losses": losses, "stats": red.stats, "buckets": len(red.buckets), "t": opt.t}
    # summed gradients of the LAST step scaled as Adam saw them, by reference-layout name
    g = {n: w.model.reference_layout(n, p.grad.detach()).cpu() / world for n, p in zip(opt.names, opt.params)}
    torch.save({"grads": g, "flat_p": opt.flat_p.cpu(), "exp_avg": opt.exp_avg.cpu()},
               os.path.join(out_dir, f"rank{rank}.pt"))
    with open(os.path.join(out_dir, f"rank{rank}.json"), "w") as f:
        json.dump(res, f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
