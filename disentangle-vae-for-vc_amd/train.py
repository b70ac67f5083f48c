"""CLI mirror of the reference's train.py for the training path (flags at /root/reference/train.py:15-46,65-71).

    python -m dvae_amd.train --train true --dataset_fp=<root> --batch-size=8 --latent-size=32 --speaker_size=4 \\
        --lr=1e-4 --epochs=10 --report-interval=5 --mse_cof=10 --kl_cof=10 --log_dir=./results

Kept flags: --batch-size --latent-size --speaker_size --lr --epochs --report-interval --mse_cof --kl_cof --seed
--dataset_fp --log_dir --train --samples_length (honoured here; the reference parses it but hard-codes 64,
train.py:53).  Parsed-and-ignored flags of the reference (--hidden-size --alpha --normalize --beta_cof --style_cof
--sample-size --no-cuda --do-not-resume --log-interval) are accepted for command-line compatibility.
--convert (voice conversion + vocoder) is out of scope.

Data parallel (new; the reference is single-device, train.py:49-58): started as one of WORLD_SIZE > 1 ranks —
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 -m dvae_amd.train ...` or with
`--gpus N` (this process then only launches N rank processes before anything touches the GPU) — every rank takes
LOCAL_RANK's GPU, joins the RCCL group, receives rank 0's weights, attaches ddp.GradReducer (bucketed all-reduce
overlapped with backward, 1/world inside Adam) and feeds from its shard of the device-resident corpus
(data.GpuPairLoader(rank=, world_size=): same epoch permutation on every rank, pairs rank, rank + world, ...).
--batch-size stays the PER-GPU batch (weak scaling).  Rank 0 writes checkpoints and logs.
"""
import argparse
import json
import os
import sys

import torch
from torch.utils.data import DataLoader


def get_parse():
    p = argparse.ArgumentParser()
    p.add_argument("--batch-size", type=int, default=2)
    p.add_argument("--hidden-size", type=str, default="400")
    p.add_argument("--speaker_size", type=int, default=4)
    p.add_argument("--latent-size", type=int, default=32)
    p.add_argument("--lr", default=1e-3, type=float)
    p.add_argument("--epochs", type=int, default=11)
    p.add_argument("--no-cuda", action="store_true", default=False)
    p.add_argument("--dataset", default="VCTK")
    p.add_argument("--seed", type=int, default=1)
    p.add_argument("--log-interval", type=int, default=500)
    p.add_argument("--report-interval", type=int, default=11)
    p.add_argument("--sample-size", type=int, default=64)
    p.add_argument("--do-not-resume", action="store_true", default=False)
    p.add_argument("--normalize", action="store_true", default=False)
    p.add_argument("--beta_cof", default=0.1, type=float)
    p.add_argument("--mse_cof", default=10, type=float)
    p.add_argument("--kl_cof", default=10, type=float)
    p.add_argument("--style_cof", default=0.1, type=float)
    p.add_argument("--samples_length", default=64, type=int)
    p.add_argument("--alpha", default=0.01, type=float)
    p.add_argument("--dataset_fp", default="/root/VCTK-Corpus/Autovc-known-speakers", type=str)
    p.add_argument("--log_dir", default="./results", type=str)
    p.add_argument("--train", type=bool, default=False)
    p.add_argument("--convert", type=bool, default=False)
    p.add_argument("--graph", type=int, default=1, help="replay the step from a hipGraph (fixed batch shape)")
    p.add_argument("--gpus", type=int, default=0,
                   help="data-parallel ranks on this node; > 1 without WORLD_SIZE in the environment: launch them")
    p.add_argument("--gpu-loader", type=int, default=-1,
                   help="1: device-resident corpus + HIP gather/crop (data.GpuPairLoader); 0: the reference's DataLoader; "
                        "-1: the GPU loader when data parallel, the DataLoader otherwise")
    return p


def get_dataset(dataset_fp, batch_size, samples_length=64, seed=None):
    from .data import SpeechDatasetGVAE
    ds = SpeechDatasetGVAE(dataset_fp, samples_length=samples_length, seed=seed)
    # drop_last keeps the batch shape fixed (hipGraph replay); the reference uses drop_last=False (train.py:55-56)
    return DataLoader(ds, batch_size=batch_size, pin_memory=True, shuffle=True, drop_last=True), ds


def launch_ranks(n, argv):
    """--gpus n without a rendezvous in the environment: n fresh rank processes of this module, started before anything
    here touches the GPU; returns the first non-zero exit code (0 if all ranks succeeded)."""
    import socket
    import subprocess
    import sys
    import time
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env["PYTHONPATH"] = root + os.pathsep + env.get("PYTHONPATH", "")
        procs.append(subprocess.Popen([sys.executable, "-c", "import dvae_amd.train as t, sys; t.main(sys.argv[1:])"]
                                      + list(argv), env=env))
    rc, live = 0, set(range(n))
    while live:
        for r in sorted(live):
            code = procs[r].poll()
            if code is None:
                continue
            live.discard(r)
            if code != 0 and rc == 0:
                rc = code
                for q in live:
                    procs[q].terminate()          # exactly the PIDs started above
        time.sleep(0.05)
    return rc if rc >= 0 else 128 - rc


def setup_data_parallel(vsc, seed):
    """Join the process group (RCCL; DVAE_DIST_BACKEND overrides for functional checks), take rank 0's weights and
    BatchNorm buffers, attach the gradient reducer.  Returns (rank, world)."""
    import torch.distributed as dist
    from . import ddp
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29519")
        backend = os.environ.get("DVAE_DIST_BACKEND", "nccl")
        if os.environ.get("DVAE_RCCL_MAX_CHANNELS"):
            os.environ["NCCL_MAX_NCHANNELS"] = os.environ["DVAE_RCCL_MAX_CHANNELS"]
        kw = {"device_id": torch.device(vsc.device)} if backend == "nccl" else {}
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    opt = vsc.optimizer
    ddp.broadcast_parameters(opt.flat_p, list(vsc.model.buffers()))
    red = ddp.GradReducer(opt.flat_g, opt.names, opt.params, opt.offsets,
                          mode=os.environ.get("DVAE_DDP_MODE", "all_reduce"),     # "rs_ag": reduce-scatter + sharded Adam
                          issue=os.environ.get("DVAE_DDP_ISSUE", "finish"))       # default: collectives after backward (nothing beside the W_hh-resident recurrences, the variant bench.py measures first and reports as the headline); "hook": issued from the backward hooks (overlap)
    red.force = os.environ.get("DVAE_FORCE_DDP", "0") == "1"      # issue the collectives with one rank too (tests)
    vsc.attach_reducer(red)
    return rank, world


def _abort_process_group():
    """Best effort, never blocks: abort the communicators of a failing rank (no collective, no barrier)."""
    try:
        import torch.distributed as dist
        if not dist.is_initialized():
            return
        pg = dist.distributed_c10d._get_default_group()
        be = None
        try:
            be = pg._get_backend(torch.device("cuda"))
        except Exception:
            pass
        for obj in (be, pg):
            for name in ("abort", "_abort", "_shutdown"):
                fn = getattr(obj, name, None)
                if callable(fn):
                    try:
                        fn()
                        return
                    except Exception:
                        pass
    except Exception:
        pass


def main(argv=None):
    args = get_parse().parse_args(argv)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        import sys
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:] if argv is None else argv))
    if args.gpus and args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    from .model.disentangled_vae import ConvolutionalMulVAE
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    n_dev = torch.cuda.device_count()
    if world > 1:
        if world > n_dev and os.environ.get("DVAE_ALLOW_SHARED_GPU", "0") != "1":
            raise SystemExit(f"{world} ranks but {n_dev} visible GPU(s): one process per GPU")
        if world > n_dev:
            from . import ops
            ops.LSTM_PERSISTENT = False      # two persistent grids cannot both be resident on one GPU
        torch.cuda.set_device(local % max(1, n_dev))
    device = torch.device("cuda", torch.cuda.current_device())
    torch.manual_seed(args.seed)
    torch.cuda.manual_seed(args.seed + 7919 * rank)      # every rank its own reparameterisation-noise stream
    if rank == 0:
        os.makedirs(args.log_dir, exist_ok=True)
        with open(os.path.join(args.log_dir, "config.json"), "w") as fp:
            json.dump(vars(args), fp, indent=4)
    vsc = ConvolutionalMulVAE(args.dataset, args.samples_length, 80, args.latent_size, args.lr, args.alpha,
                              args.log_interval, args.normalize, speaker_size=args.speaker_size, device=device,
                              latent_dim=args.latent_size, beta=args.beta_cof, batch_size=args.batch_size,
                              mse_cof=args.mse_cof, kl_cof=args.kl_cof, style_cof=args.style_cof)
    dp = world > 1 or os.environ.get("DVAE_FORCE_DDP", "0") == "1"
    if dp:
        setup_data_parallel(vsc, args.seed)
    use_gpu_loader = args.gpu_loader == 1 or (args.gpu_loader < 0 and dp)
    if use_gpu_loader:
        from .data import GpuPairLoader, SpeechDatasetGVAE
        ds = SpeechDatasetGVAE(args.dataset_fp, samples_length=args.samples_length, seed=args.seed)
        loader = GpuPairLoader(ds, args.batch_size, device=device, seed=args.seed, rank=rank, world_size=world)
    else:
        if world > 1:
            raise SystemExit("--gpu-loader 0 has no sharding: data-parallel runs feed from data.GpuPairLoader")
        loader, _ = get_dataset(args.dataset_fp, args.batch_size, args.samples_length, seed=args.seed)
    if args.graph:
        vsc.enable_graph(True)     # with a reducer attached the step still runs eagerly unless DVAE_DDP_GRAPH=1
    hist = None
    try:
        # failure injection of tests/test_hip_train_cli.py (a rank that dies while its peers are in a collective): only in
        # an explicit test mode, never from a stray variable in a production environment
        if os.environ.get("DVAE_TEST_MODE") == "1" and os.environ.get("DVAE_TEST_FAIL_RANK") == str(rank):
            raise RuntimeError(f"injected failure on rank {rank} (DVAE_TEST_MODE / DVAE_TEST_FAIL_RANK)")
        if args.train:
            hist = vsc.run_training(loader, loader, args.epochs, args.report_interval, args.sample_size,
                                    reload_model=not args.do_not_resume,
                                    checkpoints_path=os.path.join(args.log_dir, "checkpoints"),
                                    images_path=os.path.join(args.log_dir, "images"),
                                    logs_path=os.path.join(args.log_dir, "logs"),
                                    estimation_dir=os.path.join(args.log_dir, "images", "estimation"))
        if args.convert:
            raise SystemExit("--convert (mel conversion + vocoder) is outside the training hot path (SURVEY.md §8f-3)")
    except BaseException:
        # A rank that FAILS must not enter a barrier: its peers are inside an all-reduce / reduce-scatter, not a barrier, and
        # the mismatched collective would hold this rank (and its traceback) until the RCCL timeout.  Tear the group down
        # without synchronising and leave with the exception: the non-zero exit makes the launcher (launch_ranks /
        # torch.distributed.run) stop the other ranks.
        if dp:
            # the abort goes through private torch entry points whose blocking behaviour differs between versions: give it
            # five seconds on a thread of its own, then leave the hard way with the traceback already printed
            import threading
            import traceback
            th = threading.Thread(target=_abort_process_group, daemon=True)
            th.start()
            th.join(5.0)
            if th.is_alive():
                traceback.print_exc()
                sys.stderr.flush()
                os._exit(1)
        raise
    if dp:
        import torch.distributed as dist
        if dist.is_initialized():
            dist.barrier()
            dist.destroy_process_group()
    return hist


if __name__ == "__main__":
    main()
