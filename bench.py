#!/usr/bin/env python3
"""Headline benchmark: utterance pairs/sec of the disentangled-VAE TRAINING STEP
(zero_grad + forward + loss + backward + [grad all-reduce] + Adam) on synthetic [B=64, 80-mel, T=128]
per GPU, fp32, through the HIP path (BASELINE.json configs[1]; weak scaling for N > 1: B=64 per GPU).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

Rank 0 prints ONE JSON line.  Extra objects:
  roofline     dominant kernel family = the contraction kernel gemm_f32_kernel (every Linear, LSTM input projection,
               weight gradient and k5 conv), AND its single dominant instantiation (template arguments named).
               achieved = algorithmic FLOPs of the launches / their summed duration, measured HERE with HIP events
               recorded on the launch stream around every launch (dvae_prof_*) over eager steps run right after the
               graph-replayed timed region (a replayed graph carries no host hooks; same shapes, same kernels).
               peak: the default arithmetic ("fp32x3") evaluates every fp32 product as SIX exact bf16 partial products
               on the bf16 MFMA, so the matrix-pipe bound of one fp32-equivalent FLOP is 6 bf16 FLOPs: peak =
               2500 / 6 = 416.7 TFLOP/s (MI355X_MICROARCH.md: ~2.5 PF dense bf16); with --dtype fp32 (fp32 MFMA) the
               peak is 157.3 TFLOP/s; with --dtype bf16, 2500.
               traffic = fabric-side bytes per launch from the committed rocprofv3 --pmc passes
               (profiles/pmc_traffic.json, scripts/pmc_traffic.py), used only while the kernel sources still hash to
               what was profiled; otherwise null.
  cpu_baseline the CPU oracle (oracle/dvae_ref.py, the verified restatement of the reference's PyTorch-CPU step)
               timed on this box's host cores on the same workload, rank 0, N=1 only.
  other_configs  (N=1 only) BASELINE configs[2] (bf16, B=128, T=256) and the per-GPU shape of configs[4] (bf16, B=64,
               T=512), each timed here over a few graph-replayed steps, with its step-level fraction of the bf16 peak.
"""
import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3       # v_mfma_f32_32x32x2_f32 (= the vector rate)
PEAK_BF16_MFMA_TFLOPS = 2500.0     # dense bf16 MFMA (MI355X_MICROARCH.md)
PEAKS = {"fp32": PEAK_F32_MFMA_TFLOPS, "fp32x3": PEAK_BF16_MFMA_TFLOPS / 6.0, "bf16": PEAK_BF16_MFMA_TFLOPS}
ARITH = {"fp32": "fp32 operands on v_mfma_f32_32x32x2_f32",
         "fp32x3": "fp32 results on the bf16 matrix pipe: each fp32 operand split exactly into three bf16 terms, six "
                   "exact partial products per product on v_mfma_f32_32x32x16_bf16, fp32 accumulation",
         "bf16": "operands rounded to bf16, fp32 accumulation on v_mfma_f32_32x32x16_bf16 (fp32 master weights, "
                 "BatchNorm, losses, Adam)"}


def algorithmic_flops_per_pair(T):
    # 3 (fwd+bwd) * 2 segments * 2 FLOP/MAC * (28 090 368 * T + 196 608)
    return 3 * 2 * 2 * (28090368 * T + 196608)


def log(msg):
    print(f"[bench] {msg}", file=sys.stderr, flush=True)


def kernel_source_hash():
    h = hashlib.sha256()
    for f in ("gemm.hip", "common.h"):
        with open(os.path.join(ROOT, "disentangle-vae-for-vc_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


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


def physical_cores():
    """(cores usable by this process, physical cores of the host or None)."""
    try:
        usable = len(os.sched_getaffinity(0))
    except Exception:
        usable = os.cpu_count() or 1
    phys = None
    try:
        seen = set()
        pkg = core = None
        for line in open("/proc/cpuinfo"):
            if line.startswith("physical id"):
                pkg = line.split(":")[1].strip()
            elif line.startswith("core id"):
                core = line.split(":")[1].strip()
            elif not line.strip():
                if pkg is not None and core is not None:
                    seen.add((pkg, core))
                pkg = core = None
        phys = len(seen) or None
    except Exception:
        pass
    return usable, phys


def cpu_baseline(batch, frames, steps=3, timeout_s=300):
    """The oracle (kind "port") on this box's host cores, in a child process with a hard timeout so that a slow
    or oversubscribed host can never hang the benchmark.  Threads = cores usable by the process, capped at 64 (beyond
    that the oneDNN/MKL LSTM and conv kernels of this size stop scaling: measured 6.4 s/step at 64 threads of 256)."""
    import subprocess
    usable, phys = physical_cores()
    threads = max(1, min(usable, 64))
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-child", "--batch", str(batch), "--frames",
           str(frames), "--steps", str(steps), "--threads", str(threads)]
    env = dict(os.environ, HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="", OMP_NUM_THREADS=str(threads))
    base = {"unit": "utterances/sec", "cores": threads, "kind": "port", "host_logical_cpus": usable,
            "host_physical_cores": phys}
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s, env=env)
    except subprocess.TimeoutExpired:
        return dict(base, value=None, sample=f"timed out after {timeout_s}s (1 warm-up + {steps} steps of B={batch}, T={frames})")
    line = [l for l in r.stdout.splitlines() if l.startswith("CPU_BASELINE ")]
    if not line:
        return dict(base, value=None, sample="child failed: " + (r.stderr or "")[-300:])
    d = json.loads(line[-1][len("CPU_BASELINE "):])
    med = sorted(d["times"])[len(d["times"]) // 2]
    return dict(base, value=batch / med, cores=d["threads"],
                sample=f"median of {steps} full train steps (B={batch}, T={frames}, fp32, PyTorch-CPU oracle) after 1 "
                       f"warm-up; {med * 1e3:.0f} ms/step; {d['threads']} threads on a host with {usable} logical CPUs"
                       + (f" / {phys} physical cores" if phys else ""), ms_per_step=med * 1e3,
                times_ms=[round(t * 1e3) for t in d["times"]])


def build_trainer(dev, B, T, dtype, world=1, rank=0, force_ddp=False):
    import dvae_amd
    from dvae_amd import ddp, ops
    ops.set_compute_dtype(dtype)
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
    return w


def time_other_config(dev, name, B, T, dtype, steps=5):
    """One of the non-headline single-GPU configurations, timed here over a few graph-replayed steps."""
    from dvae_amd.data import SyntheticPairs
    w = build_trainer(dev, B, T, dtype)
    x1, x2, spk = SyntheticPairs(B, T, n_speakers=109, seed=4321, device=dev).batch()
    w.enable_graph(True)
    first = None
    for i in range(3):
        l = w.step(x1, x2, spk, train=True)
        first = l if first is None else first
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    last = None
    for _ in range(steps):
        last = w.step_async(x1, x2, spk)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    last = last.tolist()
    ms = 1e3 * el / steps
    fl = algorithmic_flops_per_pair(T) * B
    out = {"config": name, "dtype": "bf16" if dtype == "bf16" else "f32", "arithmetic": ARITH[dtype], "batch": B,
           "frames": T, "params": sum(p.numel() for p in w.model.parameters()), "steps": steps, "ms_per_step": ms,
           "utterances_per_sec": B * steps / el, "step_tflops_algorithmic": fl / 1e12,
           "step_frac_of_peak": fl / (ms * 1e-3) / 1e12 / PEAKS[dtype], "peak_tflops": PEAKS[dtype],
           "loss_first": first[0], "loss_last": last[0], "launch": "hipGraph replay"}
    del w
    torch.cuda.empty_cache()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=64, help="pairs per GPU")
    ap.add_argument("--frames", type=int, default=128)
    ap.add_argument("--dtype", choices=["fp32x3", "fp32", "f32", "bf16"], default="fp32x3",
                    help="arithmetic of the contractions: fp32x3 (default; fp32 results on the bf16 matrix pipe, the "
                         "headline BASELINE configs[1]), fp32 / f32 (fp32 MFMA), bf16 (configs[2]/[4] semantics; use "
                         "with --batch 128 --frames 256 for configs[2])")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true")
    ap.add_argument("--graph", type=int, default=1, help="replay the step from a captured hipGraph")
    ap.add_argument("--cpu-baseline-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--threads", type=int, default=8, help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.cpu_baseline_child:
        _cpu_baseline_child(args.batch, args.frames, args.steps, args.threads)
        return
    dtype = "fp32" if args.dtype == "f32" else args.dtype

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

    from dvae_amd import ops
    from dvae_amd.data import SyntheticPairs

    B, T = args.batch, args.frames
    w = build_trainer(dev, B, T, dtype, world, rank, force_ddp)
    data = SyntheticPairs(B, T, n_speakers=10, seed=1234 + rank, device=dev)
    x1, x2, spk = data.batch()

    def barrier():
        if world > 1 or force_ddp:
            dist.barrier()
        torch.cuda.synchronize()

    # data parallel: the RCCL collectives are captured inside the graph unless DVAE_DDP_GRAPH=0
    ddp_graph = os.environ.get("DVAE_DDP_GRAPH", "1") != "0"
    use_graph = bool(args.graph) and ((world == 1 and not force_ddp) or ddp_graph)
    if use_graph:
        w.enable_graph(True)
    log(f"rank {rank}/{world}: model built ({sum(p.numel() for p in w.model.parameters())} params), {dtype}, "
        f"warm-up x{args.warmup}")
    for _ in range(max(args.warmup, 2 if use_graph else 0)):   # graph mode: call 1 eager, call 2 captures
        w.step(x1, x2, spk, train=True)
    barrier()
    log("timed region start")
    t0 = time.perf_counter()
    last = None
    for _ in range(args.steps):
        last = w.step_async(x1, x2, spk)       # full train step; the 8 loss scalars stay on the device
    barrier()
    elapsed = time.perf_counter() - t0
    last = tuple(last.tolist())
    log(f"timed region done: {1e3 * elapsed / args.steps:.2f} ms/step")

    roof = None
    if not args.no_roofline and rank == 0 and world == 1:
        # the SAME kernels, timed with HIP events around every launch over eager steps right after the timed region
        prof_steps = min(args.steps, 5)
        w.enable_graph(False)
        ops.prof_enable(1)
        for _ in range(prof_steps):
            w.step(x1, x2, spk, train=True)
        torch.cuda.synchronize()
        tags = ops.prof_collect_tags()
        ms, launches, flops = ops.prof_collect()
        ops.prof_enable(0)
        if ms > 0 and tags:
            peak = PEAKS[dtype]
            ach = flops / (ms * 1e-3) / 1e12
            dom = tags[0]
            dom_ach = dom["flops"] / (dom["ms"] * 1e-3) / 1e12
            traffic = traffic_src = None
            try:
                with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
                    pm = json.load(f)
                if pm.get("kernel_source_sha16") == kernel_source_hash() and pm.get("workload") == f"B={B},T={T},{dtype}":
                    traffic, traffic_src = pm["gemm_f32_kernel"]["traffic_bytes_per_launch"], pm["source"]
                else:
                    traffic_src = "profiles/pmc_traffic.json was collected for other kernel sources / another workload: not used"
            except Exception:
                pass
            roof = {"bound": "mfma", "kernel": "contraction family: gemm_f32_kernel<...> + gemm_x3_tall_kernel<...> (" + ARITH[dtype] + ")",
                    "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak,
                    "peak_note": {"fp32x3": "2500 TFLOP/s dense bf16 MFMA / 6 bf16 partial products per fp32 product",
                                  "fp32": "fp32 MFMA = vector rate", "bf16": "dense bf16 MFMA"}[dtype],
                    "traffic": traffic, "traffic_unit": "bytes per launch (L2<->fabric, PMC)", "traffic_source": traffic_src,
                    "algorithmic_bytes_per_launch": sum(t["bytes"] for t in tags) / max(1, launches),
                    "launches_per_step": launches / prof_steps, "kernel_ms_per_step": ms / prof_steps,
                    "avg_launch_us": 1e3 * ms / max(1, launches), "flops_per_step": flops / prof_steps,
                    "dominant_instantiation": {"kernel": dom["kernel"], "achieved": dom_ach, "frac": dom_ach / peak,
                                               "launches_per_step": dom["launches"] / prof_steps,
                                               "kernel_ms_per_step": dom["ms"] / prof_steps,
                                               "avg_launch_us": 1e3 * dom["ms"] / max(1, dom["launches"]),
                                               "algorithmic_bytes_per_launch": dom["bytes"] / max(1, dom["launches"])},
                    "instantiations": [{"kernel": t["kernel"], "ms_per_step": t["ms"] / prof_steps,
                                        "launches_per_step": t["launches"] / prof_steps,
                                        "tflops": t["flops"] / (t["ms"] * 1e-3) / 1e12} for t in tags[:6]],
                    "timed": f"HIP events around every launch, {prof_steps} eager steps right after the graph-replayed "
                             "timed region"}
    if world > 1 or force_ddp:
        tmax = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    final_line = None
    if rank == 0:
        ms_step = 1e3 * elapsed / args.steps
        value = world * B * args.steps / elapsed
        step_flops = algorithmic_flops_per_pair(T) * B
        n_params = sum(p.numel() for p in w.model.parameters())
        cfg_name = "configs[1]" if (B, T, dtype) in ((64, 128, "fp32x3"), (64, 128, "fp32")) else "custom"
        if world > 1 and B == 64 and T == 128:
            cfg_name = "configs[3]-style (weak scaling: B=64 per GPU)"
        out = {"metric": f"utterances/sec (B={B}, 80-mel, T={T}) train step", "value": value, "unit": "utterances/sec",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_step,
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": "bf16" if dtype == "bf16" else "f32",
               "data": "synthetic U[0,1) mel pairs, random-init weights",
               "config": {"workload": f"{cfg_name}: train step, B={B} pairs/GPU, 80-mel, T={T}, 10 synthetic speakers, "
                                      "speaker_size=4, latent=32, Adam lr=1e-4; fp32 tensors, master weights, BatchNorm, "
                                      "losses and Adam",
                          "arithmetic": ARITH[dtype], "compute_mode": dtype,
                          "global_batch": world * B, "frames": T, "parallelism": f"dp{world}",
                          "launch": ("hipGraph replay" + (" (RCCL all-reduce captured in the graph)" if world > 1 else ""))
                          if use_graph else "eager", "params": n_params},
               "step_tflops_algorithmic": step_flops / 1e12,
               "step_frac_of_peak": step_flops / (ms_step * 1e-3) / 1e12 / PEAKS[dtype],
               "final_loss": last[0] if last else None}
        if roof:
            out["roofline"] = roof
        del w
        torch.cuda.empty_cache()
        if world == 1 and not force_ddp and not args.no_other_configs and (B, T) == (64, 128):
            others = []
            for name, b, t in (("configs[2]: 1xMI355X bf16, B=128, T=256", 128, 256),
                               ("configs[4] per-GPU shape: bf16, B=64, T=512", 64, 512)):
                try:
                    log(f"timing {name}")
                    others.append(time_other_config(dev, name, b, t, "bf16"))
                except Exception as e:      # never lose the headline line to a side measurement
                    others.append({"config": name, "error": repr(e)[:300]})
            out["other_configs"] = others
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
