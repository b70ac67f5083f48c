#!/usr/bin/env python3
"""Headline benchmark: utterance pairs/sec of the disentangled-VAE TRAINING STEP
(zero_grad + forward + loss + backward + [grad all-reduce] + Adam) on synthetic [B=64, 80-mel, T=128]
per GPU, fp32, through the HIP path (BASELINE.json configs[1]; weak scaling for N > 1: B=64 per GPU).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

Rank 0 prints ONE JSON line.  Extra objects:
  roofline     dominant kernel family = the fp32-MFMA contraction kernel (gemm_f32_kernel: every Linear, LSTM input
               projection, weight gradient and k5 conv).  achieved = algorithmic FLOPs of its launches / their summed
               duration, measured with HIP events recorded on the launch stream around every launch INSIDE the
               timed region (dvae_prof_*); peak = 157.3 TFLOP/s fp32 MFMA (MI355X_MICROARCH.md).
  cpu_baseline the CPU oracle (oracle/dvae_ref.py, the verified restatement of the reference's PyTorch-CPU step)
               timed on this box's host cores on the same workload, rank 0, N=1 only.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3
PEAK_BF16_MFMA_TFLOPS = 2500.0     # dense bf16 MFMA (MI355X_MICROARCH.md)


def algorithmic_flops_per_pair(T):
    # 3 (fwd+bwd) * 2 segments * 2 FLOP/MAC * (28 090 368 * T + 196 608)
    return 3 * 2 * 2 * (28090368 * T + 196608)


def log(msg):
    print(f"[bench] {msg}", file=sys.stderr, flush=True)


def _cpu_baseline_child(batch, frames, steps, threads):
    """Runs in a CHILD process (never touches the GPU): times the CPU oracle's train step."""
    torch.set_num_threads(threads)
    from oracle.dvae_ref import RefTrainer
    from oracle.fill import synthetic_pair
    tr = RefTrainer(batch, n_frames=frames)
    x1, x2 = synthetic_pair(batch, frames, 1234)
    g = torch.Generator().manual_seed(0)
    tr.step(x1, x2, tr.draw_eps(batch, g), train=True)   # warm-up
    ts = []
    for _ in range(steps):
        t0 = time.perf_counter()
        tr.step(x1, x2, tr.draw_eps(batch, g), train=True)
        ts.append(time.perf_counter() - t0)
    print("CPU_BASELINE " + json.dumps({"times": ts, "threads": torch.get_num_threads()}), flush=True)


def cpu_baseline(batch, frames, steps=2, timeout_s=240):
    """The oracle (kind "port") on this box's host cores, in a child process with a hard timeout so that a slow
    or oversubscribed host can never hang the benchmark.  Threads = physical cores available to the process,
    capped at 64 (beyond that the oneDNN/MKL LSTM and conv kernels of this size stop scaling)."""
    import subprocess
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        cores = os.cpu_count() or 1
    threads = max(1, min(cores, 64))
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-child", "--batch", str(batch), "--frames",
           str(frames), "--steps", str(steps), "--threads", str(threads)]
    env = dict(os.environ, HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="", OMP_NUM_THREADS=str(threads))
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s, env=env)
    except subprocess.TimeoutExpired:
        return {"value": None, "unit": "utterances/sec", "cores": threads, "kind": "port",
                "sample": f"timed out after {timeout_s}s (1 warm-up + {steps} steps of B={batch}, T={frames})"}
    line = [l for l in r.stdout.splitlines() if l.startswith("CPU_BASELINE ")]
    if not line:
        return {"value": None, "unit": "utterances/sec", "cores": threads, "kind": "port",
                "sample": "child failed: " + (r.stderr or "")[-300:]}
    d = json.loads(line[-1][len("CPU_BASELINE "):])
    best = sorted(d["times"])[len(d["times"]) // 2]
    return {"value": batch / best, "unit": "utterances/sec", "cores": d["threads"], "kind": "port",
            "sample": f"median of {steps} full train steps (B={batch}, T={frames}, fp32, PyTorch-CPU oracle) after "
                      f"1 warm-up; {best * 1e3:.0f} ms/step; host has {cores} cores", "ms_per_step": best * 1e3}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=64, help="pairs per GPU")
    ap.add_argument("--frames", type=int, default=128)
    ap.add_argument("--dtype", choices=["f32", "bf16"], default="f32",
                    help="compute mode of the contractions; f32 = BASELINE configs[1] (the headline), bf16 = "
                         "configs[2]/[4] semantics (use with --batch 128 --frames 256 for configs[2])")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--graph", type=int, default=1, help="replay the step from a captured hipGraph (N=1 only)")
    ap.add_argument("--cpu-baseline-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--threads", type=int, default=8, help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.cpu_baseline_child:
        _cpu_baseline_child(args.batch, args.frames, args.steps, args.threads)
        return

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    force_ddp = os.environ.get("DVAE_FORCE_DDP", "0") == "1"      # exercise the RCCL path with one rank (testing)
    if world > 1 or force_ddp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)

    import dvae_amd
    from dvae_amd import ddp, ops
    from dvae_amd.data import SyntheticPairs

    B, T = args.batch, args.frames
    bf16 = args.dtype == "bf16"
    if bf16:
        ops.set_compute_dtype("bf16")
    torch.manual_seed(1234)
    w = dvae_amd.ConvolutionalMulVAE("VCTK", T, 80, 32, 1e-4, 0.01, 500, False, batch_size=B, speaker_size=4,
                                     device=dev, latent_dim=32, mse_cof=10, kl_cof=10)
    w.model.train()
    # identical weights on every rank (same seed, then broadcast), but an INDEPENDENT reparameterisation-noise stream
    # per rank: the global batch of a data-parallel step must not carry one rank's eps eight times
    torch.cuda.manual_seed(1234 + 7919 * rank)
    if world > 1 or force_ddp:
        ddp.broadcast_parameters(w.optimizer.flat_p, [b for b in w.model.buffers()])
        red = ddp.GradReducer(w.optimizer.flat_g, w.optimizer.names, w.optimizer.params, w.optimizer.offsets)
        red.force = force_ddp
        w.attach_reducer(red)
    data = SyntheticPairs(B, T, n_speakers=10, seed=1234 + rank, device=dev)
    x1, x2, spk = data.batch()

    def barrier():
        if world > 1 or force_ddp:
            dist.barrier()
        torch.cuda.synchronize()

    # data parallel: the RCCL collectives are captured inside the graph unless DVAE_DDP_GRAPH=0
    use_graph = bool(args.graph) and ((world == 1 and not force_ddp) or os.environ.get("DVAE_DDP_GRAPH", "1") != "0")
    if use_graph:
        w.enable_graph(True)
    log(f"rank {rank}/{world}: model built ({sum(p.numel() for p in w.model.parameters())} params), warm-up x{args.warmup}")
    for _ in range(max(args.warmup, 2 if use_graph else 0)):   # graph mode: call 1 eager, call 2 captures
        w.step(x1, x2, spk, train=True)
    barrier()
    log("timed region start")
    prof_live = (not args.no_roofline) and not use_graph
    if prof_live:
        ops.prof_enable(1)
    t0 = time.perf_counter()
    last = None
    for _ in range(args.steps):
        last = w.step_async(x1, x2, spk)       # full train step; the 8 loss scalars stay on the device
    barrier()
    elapsed = time.perf_counter() - t0
    last = tuple(last.tolist())
    log(f"timed region done: {1e3 * elapsed / args.steps:.2f} ms/step")
    roof = None
    if not args.no_roofline:
        prof_steps = args.steps
        if not prof_live:
            # graph replays carry no host-side hooks: time the SAME kernels with HIP events over eager steps
            # run right after the timed region (identical shapes and launches; `value` is unaffected)
            prof_steps = min(args.steps, 5)
            w.enable_graph(False)
            ops.prof_enable(1)
            for _ in range(prof_steps):
                w.step(x1, x2, spk, train=True)
            torch.cuda.synchronize()
        ms, launches, flops = ops.prof_collect()
        ops.prof_enable(0)
        traffic, traffic_src, traffic_alg = None, None, None
        try:   # HBM-side bytes per launch of this kernel family from the committed PMC passes (not collectable live)
            with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
                pm = json.load(f)
            if B == 64 and T == 128 and not bf16:
                traffic, traffic_src = pm["gemm_f32_kernel"]["traffic_bytes_per_launch"], pm["source"]
                traffic_alg = pm["gemm_f32_kernel"].get("algorithmic_bytes_per_launch")
        except Exception:
            pass
        if ms > 0:
            ach = flops / (ms * 1e-3) / 1e12
            peak = PEAK_BF16_MFMA_TFLOPS if bf16 else PEAK_F32_MFMA_TFLOPS
            roof = {"bound": "mfma", "kernel": "gemm_f32_kernel (" + ("v_mfma_f32_32x32x16_bf16, fp32 tensors in HBM"
                                                                       if bf16 else "v_mfma_f32_32x32x2_f32") + ")",
                    "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak,
                    "traffic": traffic, "traffic_unit": "bytes per launch (L2<->fabric, PMC)",
                    "traffic_source": traffic_src, "algorithmic_bytes_per_launch": traffic_alg, "launches_per_step": launches / prof_steps,
                    "kernel_ms_per_step": ms / prof_steps, "avg_launch_us": 1e3 * ms / max(1, launches),
                    "flops_per_step": flops / prof_steps,
                    "timed": "HIP events around every launch, " + ("inside the timed region" if prof_live else
                             f"{prof_steps} eager steps right after the graph-replayed timed region")}
    if world > 1 or force_ddp:
        tmax = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    final_line = None
    if rank == 0:
        ms_step = 1e3 * elapsed / args.steps
        value = world * B * args.steps / elapsed
        step_flops = algorithmic_flops_per_pair(T) * B
        out = {"metric": f"utterances/sec (B={B}, 80-mel, T={T}) train step", "value": value, "unit": "utterances/sec",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_step,
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype,
               "data": "synthetic U[0,1) mel pairs, random-init weights",
               "config": {"workload": (f"configs[2]-style: bf16-compute train step (bf16 MFMA operands, fp32 accumulate, "
                                       f"fp32 tensors / master weights / Adam), B={B} pairs/GPU, 80-mel, T={T}, "
                                       if bf16 else f"configs[1]: fp32 train step, B={B} pairs/GPU, 80-mel, T={T}, ")
                                      + "10 synthetic speakers, speaker_size=4, latent=32, Adam lr=1e-4",
                          "global_batch": world * B, "frames": T, "parallelism": f"dp{world}",
                          "launch": "hipGraph replay" if use_graph else "eager",
                          "params": sum(p.numel() for p in w.model.parameters())},
               "step_tflops_algorithmic": step_flops / 1e12,
               "step_frac_of_fp32_mfma_peak": step_flops / (ms_step * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS,
               "final_loss": last[0] if last else None}
        if roof:
            out["roofline"] = roof
        if world == 1 and not args.no_cpu_baseline:
            log("timing the CPU oracle on the host cores (child process, bounded)")
            cb = cpu_baseline(B, T)
            out["cpu_baseline"] = cb
            if cb.get("value"):
                out["speedup_vs_cpu_baseline"] = value / cb["value"]
        final_line = json.dumps(out)
    if world > 1 or force_ddp:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        # RCCL writes a banner through C stdio, which sits in libc's buffer until exit: flush it first so that the
        # JSON line is the LAST line of stdout
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(final_line, flush=True)


if __name__ == "__main__":
    main()
