"""Child process of tests/test_hip_ddp.py: ONE rank on cuda:0 with the real RCCL backend ("nccl"), so that the
data-parallel step — ops.grad_ready_hook -> GradReducer (bucketed all-reduce issued from the autograd thread while the
HIP backward kernels are still being enqueued) -> FlatAdam(grad_scale) — runs on the HIP kernels.  Started as a FRESH
process: nothing has touched the GPU before init_process_group.  Prints one JSON line.

argv: <graph 0|1> <lr> <steps>"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    use_graph, lr, steps = int(sys.argv[1]), float(sys.argv[2]), int(sys.argv[3])
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", device_id=dev, rank=0, world_size=1)
    import dvae_amd
    from dvae_amd import ddp
    from oracle.fill import fill_state_dict, synthetic_eps, synthetic_pair     # deterministic weights / inputs only

    B, T = 4, 64

    def make(with_reducer):
        w = dvae_amd.ConvolutionalMulVAE("VCTK", T, 80, 32, lr, 0.01, 500, False, batch_size=B, speaker_size=4,
                                         device=dev, latent_dim=32, mse_cof=10, kl_cof=10)
        w.model.load_state_dict(fill_state_dict(w.model.state_dict()))
        w.model.train()
        red = None
        if with_reducer:
            ddp.broadcast_parameters(w.optimizer.flat_p, list(w.model.buffers()))
            red = ddp.GradReducer(w.optimizer.flat_g, w.optimizer.names, w.optimizer.params, w.optimizer.offsets,
                                  mode=os.environ.get("DVAE_DDP_MODE", "all_reduce"),
                                  issue=os.environ.get("DVAE_DDP_ISSUE", "hook"))
            red.force = True                    # issue the collectives although world_size == 1
            w.attach_reducer(red)
            if use_graph:
                w.enable_graph(True, ddp=True)
        return w, red

    a, _ = make(False)
    b, red = make(True)
    out = {"graph": use_graph, "lr": lr, "buckets": len(red.buckets), "losses_plain": [], "losses_ddp": []}
    for i in range(steps):
        x1, x2 = (t.cuda() for t in synthetic_pair(B, T, 100 + i))
        eps = synthetic_eps(B, seed=200 + i)
        a.model.eps_override = eps
        b.model.eps_override = eps
        out["losses_plain"].append(list(a.step(x1, x2, None, train=True)))
        out["losses_ddp"].append(list(b.step(x1, x2, None, train=True)))
    torch.cuda.synchronize()
    from dvae_amd import ops
    out["deterministic"] = bool(ops.deterministic())
    out["ddp_mode"] = red.mode
    out["graph_captured"] = b._graph is not None
    out["stats"] = red.stats
    out["views_intact"] = bool(a.optimizer.views_intact() and b.optimizer.views_intact())
    out["t"] = [a.optimizer.t, b.optimizer.t]
    # distance of the two parameter vectors (Adam's first updates are lr * sign(g): every gradient element that is zero
    # up to round-off moves its weight by +-lr independently in the two trainers, so the trajectories are not bitwise equal)
    out["param_dist_rel"] = float((a.optimizer.flat_p - b.optimizer.flat_p).double().norm() /
                                  a.optimizer.flat_p.double().norm())
    out["param_moved_rel"] = float(lr * steps * (a.optimizer.numel ** 0.5) / a.optimizer.flat_p.double().norm())
    ma, mb = a.optimizer.exp_avg, b.optimizer.exp_avg
    out["exp_avg_rel"] = float((ma - mb).norm() / ma.norm())
    print("DDPCHILD " + json.dumps(out), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
