#!/usr/bin/env python3
"""Headline benchmark: utterance pairs/sec of the disentangled-VAE TRAINING STEP
(zero_grad + forward + loss + backward + [grad all-reduce] + Adam) on synthetic [B=64, 80-mel, T=128]
per GPU, fp32, through the HIP path (BASELINE.json configs[1]; weak scaling for N > 1: B=64 per GPU).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

`--gpus N` MEANS N ranks.  Started without a rendezvous in the environment (no WORLD_SIZE) and N > 1, this process is
only a LAUNCHER: before anything touches the GPU it starts N fresh rank processes of this script (RANK / LOCAL_RANK /
WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT set, one GPU each), relays rank 0's stdout — its last line is the JSON
line — and exits with the first non-zero exit code of any rank (the others are terminated by PID).  Started by
torch.distributed.run (WORLD_SIZE set), it is one of the ranks; WORLD_SIZE != --gpus is an error, never a silent N=1 run.
`--dry-launch` runs the same launcher / rendezvous / barrier / max-over-ranks timing with the gloo backend and NO GPU work
(tests/test_host_logic.py: the launcher is exercised on CPU).

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
  roofline_lstm  the LSTM recurrence family (frame launches + the W_hh-resident persistent launches), timed the same way:
               achieved TFLOP/s against the matrix-pipe peak of the arithmetic each instantiation runs, and the
               ALGORITHMIC L2 -> CU operand bytes of its tiling per second against the 17-18.8 TB/s the XCD L2s
               deliver (MI355X_MICROARCH.md: 66-73 GB/s per CU).
  ms_per_step_sync  the same step through `step()`: + the device->host copy of the 8 loss scalars every step (the
               reference's `step` ends in 8 .item() syncs, variational_base_vae.py:70); `ms_per_step` is `step_async`.
  cpu_baseline the CPU oracle (oracle/dvae_ref.py, the verified restatement of the reference's PyTorch-CPU step)
               timed on this box's host cores on the same workload, rank 0, N=1 only: once with 64 threads and once
               with all physical cores; `value` is the FASTER of the two, both samples listed.
  N > 1        the EAGER data-parallel step, timed once per exchange variant — bucketed all-reduce or reduce-scatter +
               sharded Adam + all-gather, collectives issued from the backward hooks or after backward — each over exactly
               --steps steps (ddp_variants_ms_per_step); the headline is the fastest (no multi-GPU box has been available to
               choose in advance).  The step with the collectives captured inside the hipGraph is attempted only with
               DVAE_BENCH_DDP_GRAPH=1.  Diagnostics: rccl_ranks, visible devices, per-rank ms, buckets,
               allreduce_exposed_ms (step minus the same step with the reducer detached).  The line carries the SAME objects
               as the N = 1 line: roofline / roofline_lstm (rank 0 alone, reducer detached, after the timed variants),
               cpu_baseline (after the process group is gone) and scaling_vs_n1 (value / the single-rank graph-replayed step
               timed in the same processes, all ranks at once; DVAE_BENCH_N1_MS supplies a number measured elsewhere).  A
               watchdog over the exchange variants (DVAE_BENCH_VARIANT_TIMEOUT) prints the marked line of what was measured
               if one of them wedges — and then every rank exits 5.
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
    for f in ("gemm.hip", "gemm256.hip", "gemm_common.h", "common.h"):
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


def _cpu_baseline_run(batch, frames, steps, threads, timeout_s):
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-child", "--batch", str(batch), "--frames",
           str(frames), "--steps", str(steps), "--threads", str(threads)]
    env = dict(os.environ, HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="", OMP_NUM_THREADS=str(threads))
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s, env=env)
    except subprocess.TimeoutExpired:
        return {"threads": threads, "error": f"timed out after {timeout_s}s"}
    line = [l for l in r.stdout.splitlines() if l.startswith("CPU_BASELINE ")]
    if not line:
        return {"threads": threads, "error": "child failed: " + (r.stderr or "")[-300:]}
    d = json.loads(line[-1][len("CPU_BASELINE "):])
    med = sorted(d["times"])[len(d["times"]) // 2]
    return {"threads": d["threads"], "ms_per_step": med * 1e3, "value": batch / med,
            "times_ms": [round(t * 1e3) for t in d["times"]]}


def cpu_baseline(batch, frames, steps=3, timeout_s=240):
    """The oracle (kind "port") on this box's host cores, in child processes with a hard timeout each (a slow or
    oversubscribed host can never hang the benchmark).  Two samples — 64 threads (where the oneDNN/MKL LSTM and conv
    kernels of this size stopped scaling on the boxes measured so far) and ALL physical cores — and the
    faster one is the baseline."""
    usable, phys = physical_cores()
    counts = sorted({max(1, min(usable, 64)), max(1, min(usable, phys or usable))})
    # the all-cores sample is there to SHOW that it is the slower one (3x on a 128-core host): one timed step is enough
    samples = [_cpu_baseline_run(batch, frames, steps if i == 0 else 1, n, timeout_s) for i, n in enumerate(counts)]
    base = {"unit": "utterances/sec", "kind": "port", "host_logical_cpus": usable, "host_physical_cores": phys,
            "samples": samples}
    good = [x for x in samples if x.get("value")]
    if not good:
        return dict(base, value=None, cores=counts[-1], sample="no sample completed: " + json.dumps(samples)[:300])
    best = max(good, key=lambda x: x["value"])
    return dict(base, value=best["value"], cores=best["threads"], ms_per_step=best["ms_per_step"],
                sample=f"median of {steps} full train steps (1 for the second sample) (B={batch}, T={frames}, fp32, PyTorch-CPU "
                       f"oracle) after 1 warm-up, at {' and '.join(str(x['threads']) for x in samples)} threads on a host with {usable} "
                       f"logical CPUs" + (f" / {phys} physical cores" if phys else "") +
                       f"; the faster sample ({best['threads']} threads, {best['ms_per_step']:.0f} ms/step) is the baseline")


def _free_port():
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def visible_gpu_count():
    """GPUs this process's children will see, WITHOUT touching the HIP runtime (the launcher never does): the KFD
    topology's nodes with SIMDs (CPUs have simd_count 0), cut down by HIP_/ROCR_/CUDA_VISIBLE_DEVICES lists.  None when
    the topology cannot be read (then --gpus is trusted and the ranks themselves check)."""
    import glob
    n = 0
    paths = glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties")
    if not paths:
        return None
    for path in paths:
        try:
            for line in open(path):
                k, _, v = line.partition(" ")
                if k == "simd_count" and int(v) > 0:
                    n += 1
                    break
        except OSError:
            return None
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([x for x in v.split(",") if x.strip() != ""]))
    return n


def launch_ranks(n, argv, dry=False):
    """`--gpus n` without a rendezvous in the environment: start n fresh rank processes of this script (one per GPU) and
    relay rank 0's stdout.  Runs BEFORE anything in this process touches the GPU — the device count comes from the KFD
    topology in sysfs (visible_gpu_count), not from the HIP runtime — and never replaces this process with another
    program.  Returns the exit code: 0 only if every rank exited 0."""
    import subprocess
    import threading
    if not dry:
        n_dev = visible_gpu_count()
        if n_dev is not None and n_dev < n and os.environ.get("DVAE_ALLOW_SHARED_GPU", "0") != "1":
            log(f"--gpus {n} but only {n_dev} GPU(s) are visible: one process per GPU is the design")
            return 2
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: what RCCL needs on this pool
        env.setdefault("OMP_NUM_THREADS", "4")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, text=(r == 0)))
    log(f"launcher: started {n} rank processes (pids {[p.pid for p in procs]}), rendezvous 127.0.0.1:{port}")
    lines = []

    def pump():
        for line in procs[0].stdout:
            lines.append(line)
    th = threading.Thread(target=pump, daemon=True)
    th.start()
    rc = 0
    live = set(range(n))
    while live:
        for r in sorted(live):
            code = procs[r].poll()
            if code is None:
                continue
            live.discard(r)
            if code != 0 and rc == 0:
                rc = code
                log(f"launcher: rank {r} exited with {code}; stopping the others")
                for q in live:
                    procs[q].terminate()              # exactly the PIDs started above
        time.sleep(0.05)
    th.join(10)
    sys.stdout.write("".join(lines))
    sys.stdout.flush()
    if rc == 0 and not any(l.lstrip().startswith("{") for l in lines):
        log("launcher: rank 0 printed no JSON line")
        rc = 4
    return rc if rc >= 0 else 128 - rc


def dry_launch(args, world, rank):
    """The rendezvous, barrier and max-over-ranks timing of a real run with the gloo backend and no GPU work: proves that
    `--gpus N` reaches N cooperating ranks (CPU test)."""
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29517")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    if dist.get_world_size() != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the process group has {dist.get_world_size()} ranks")
    one = torch.ones(1)
    dist.all_reduce(one)
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(0.001 * (1 + rank))               # ranks differ: the line must carry the SLOWEST rank's time
    dist.barrier()
    own = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    dist.all_reduce(own, op=dist.ReduceOp.MAX)
    if rank == 0:
        el = float(own.item())
        print(json.dumps({"metric": f"utterances/sec (B={args.batch}, 80-mel, T={args.frames}) train step", "value": None,
                          "unit": "utterances/sec", "n_gpus": int(one.item()), "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": 1e3 * el / max(1, args.steps), "higher_is_better": True, "scaling": "weak",
                          "vs_baseline": None, "dtype": "f32", "data": "none", "dry_launch": True,
                          # the keys of a real N > 1 line (tests/test_launcher.py holds them): the cooperating ranks, the
                          # DEFAULT exchange (train.py's: the headline) and the fastest one beside it
                          "rccl_ranks": dist.get_world_size(), "backend": "gloo",
                          "ddp_default": {"variant": f"{os.environ.get('DVAE_DDP_MODE', 'all_reduce')}:"
                                                     f"{os.environ.get('DVAE_DDP_ISSUE', 'finish')}",
                                          "ms_per_step": 1e3 * el / max(1, args.steps), "value": None},
                          "ddp_fastest": {"variant": None, "ms_per_step": None, "value": None},
                          "ddp_variants_ms_per_step": {},
                          "config": {"workload": "dry launch: rendezvous + barrier only, no GPU work",
                                     "parallelism": f"dp{dist.get_world_size()}"}}), flush=True)
    dist.barrier()
    dist.destroy_process_group()


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
        red = ddp.GradReducer(w.optimizer.flat_g, w.optimizer.names, w.optimizer.params, w.optimizer.offsets,
                              mode=os.environ.get("DVAE_DDP_MODE", "all_reduce"),      # or "rs_ag": sharded Adam
                              # measured FIRST: the form with nothing beside the W_hh-resident recurrences ("finish":
                              # collectives after backward) — should an overlapped variant wedge later, the watchdog has
                              # this one's line; the overlapped form ("hook") is one of the variants timed after it
                              issue=os.environ.get("DVAE_DDP_ISSUE", "finish"))
        red.force = force_ddp
        w.attach_reducer(red)
    return w


WATCHDOG_RC = 5      # exit code of every rank when the watchdog over the exchange variants fired (a wedged collective)
L2_CU_TBPS = (17.0, 18.8)     # XCD L2 -> CU delivery, chip-wide (256 CUs x 66-73 GB/s, MI355X_MICROARCH.md)
LSTM_KINDS = {0: "frame launches, forward", 1: "frame launches, backward", 2: "W_hh-resident persistent launch, forward",
              3: "W_hh-resident persistent launch, backward", 4: "H=64 whole-sequence launch, forward",
              5: "H=64 whole-sequence launch, backward"}
LSTM_PM = {0: ("fp32", PEAK_F32_MFMA_TFLOPS), 1: ("bf16", PEAK_BF16_MFMA_TFLOPS), 2: ("fp32x3", PEAK_BF16_MFMA_TFLOPS / 6.0)}


_XCD_LOCAL_PER_STEP = None


def profile_families(w, x1, x2, spk, ops, prof_steps):
    """Eager steps with HIP events around every launch of one kernel family (the library's dvae_prof_* hooks): family 1
    the contraction kernels, family 2 the LSTM recurrence.  Returns (tags1, (ms, launches, flops)1, tags2, (...)2)."""
    out = []
    w.enable_graph(False)
    global _XCD_LOCAL_PER_STEP
    for fam in (1, 2):
        ops.prof_enable(fam)
        n_loc = ops.lstm_pers_local_launches() if fam == 2 else 0
        for _ in range(prof_steps):
            w.step(x1, x2, spk, train=True)
        torch.cuda.synchronize()
        if fam == 2:      # persistent launches of these steps whose hand-offs stayed inside one XCD's L2
            _XCD_LOCAL_PER_STEP = (ops.lstm_pers_local_launches() - n_loc) / prof_steps
        tags = ops.prof_collect_tags()
        tot = ops.prof_collect()
        ops.prof_enable(0)
        out += [tags, tot]
    return out


def lstm_roofline(tags, tot, prof_steps):
    ms, launches, flops = tot
    if ms <= 0 or not tags:
        return None
    inst, peak_w = [], 0.0
    for t in tags:
        kind, hc, pm = t["tag"] & 15, (t["tag"] >> 4) & 15, (t["tag"] >> 8) & 3
        name, peak = LSTM_PM[pm]
        tf = t["flops"] / (t["ms"] * 1e-3) / 1e12
        peak_w += t["ms"] * peak
        inst.append({"kernel": f"{LSTM_KINDS.get(kind, kind)}, H={ {0: 64, 1: 512, 2: 1024}.get(hc, '?') }, {name} recurrent product",
                     "ms_per_step": t["ms"] / prof_steps, "calls_per_step": t["launches"] / prof_steps, "tflops": tf,
                     "frac_of_mfma_peak": tf / peak,
                     "l2_to_cu_algorithmic_tb_per_s": (t["bytes"] / (t["ms"] * 1e-3) / 1e12) if t["bytes"] else None})
    ach = flops / (ms * 1e-3) / 1e12
    peak = peak_w / ms                      # time-weighted matrix-pipe peak of the arithmetics the family ran
    by = sum(t["bytes"] for t in tags)
    tbps = by / (ms * 1e-3) / 1e12
    return {"bound": "l2 (XCD L2 -> CU operand streaming) / cross-workgroup hand-off latency for the persistent launches",
            "kernel": "LSTM recurrence family: lstm_step_*_v5 (one launch per frame), lstm_pers_* (one launch per sequence), "
                      "lstm_seq_*_h64", "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak,
            "peak_note": "time-weighted over the family: 416.7 (fp32x3), 157.3 (fp32 MFMA) or 2500 (bf16) per instantiation",
            "l2_to_cu_algorithmic_bytes_per_step": by / prof_steps, "l2_to_cu_achieved_tb_per_s": tbps,
            "l2_to_cu_peak_tb_per_s": list(L2_CU_TBPS), "l2_to_cu_frac": tbps / L2_CU_TBPS[0],
            "kernel_ms_per_step": ms / prof_steps, "calls_per_step": launches / prof_steps, "flops_per_step": flops / prof_steps,
            "instantiations": inst,
            "xcd_local_launches_per_step": _XCD_LOCAL_PER_STEP,      # of the persistent launches: hand-offs kept in one XCD's L2
            "timed": f"HIP events around every recurrence call, {prof_steps} eager steps after the timed region"}


def time_other_config(dev, name, B, T, dtype, steps=15):
    """One of the non-headline single-GPU configurations, timed here over a few graph-replayed steps, with the
    achieved rates of its two dominant kernel families (eager steps with HIP events afterwards)."""
    from dvae_amd import ops
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
    ops.lstm_pers_check()
    last = last.tolist()
    ms = 1e3 * el / steps
    fl = algorithmic_flops_per_pair(T) * B
    out = {"config": name, "dtype": "bf16" if dtype == "bf16" else "f32", "arithmetic": ARITH[dtype], "batch": B,
           "frames": T, "params": sum(p.numel() for p in w.model.parameters()), "steps": steps, "ms_per_step": ms,
           "utterances_per_sec": B * steps / el, "step_tflops_algorithmic": fl / 1e12,
           "step_frac_of_peak": fl / (ms * 1e-3) / 1e12 / PEAKS[dtype], "peak_tflops": PEAKS[dtype],
           "loss_first": first[0], "loss_last": last[0], "launch": "hipGraph replay"}
    try:
        t1, tot1, t2, tot2 = profile_families(w, x1, x2, spk, ops, 2)
        if tot1[0] > 0:
            ach = tot1[2] / (tot1[0] * 1e-3) / 1e12
            out["roofline"] = {"bound": "mfma", "kernel": "contraction family (gemm_bf16_256_kernel / gemm_bf16_tall_kernel / gemm_f32_kernel<MODE=1>)",
                               "achieved": ach, "peak": PEAKS[dtype], "unit": "TFLOP/s", "frac": ach / PEAKS[dtype],
                               "kernel_ms_per_step": tot1[0] / 2, "launches_per_step": tot1[1] / 2,
                               "instantiations": [{"kernel": t["kernel"], "ms_per_step": t["ms"] / 2,
                                                   "tflops": t["flops"] / (t["ms"] * 1e-3) / 1e12} for t in t1[:4]]}
        rl = lstm_roofline(t2, tot2, 2)
        if rl:
            out["roofline_lstm"] = rl
    except Exception as e:
        out["roofline_error"] = repr(e)[:200]
    del w
    torch.cuda.empty_cache()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)      # 50 x 18 ms: a timed region of ~1 s (round 5 review: 0.35 s was short)
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
    ap.add_argument("--dry-launch", action="store_true",
                    help="launcher / rendezvous check only: gloo backend, no GPU work, the line carries dry_launch=true")
    ap.add_argument("--cpu-baseline-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--threads", type=int, default=8, help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.cpu_baseline_child:
        _cpu_baseline_child(args.batch, args.frames, args.steps, args.threads)
        return
    dtype = "fp32" if args.dtype == "f32" else args.dtype
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # launcher: nothing in THIS process ever touches the GPU
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:], dry=args.dry_launch))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the environment says WORLD_SIZE={world}: refusing to run a "
                         f"{world}-rank job under an n_gpus={args.gpus} label (start it with --gpus {world}, or unset "
                         "WORLD_SIZE and let bench.py launch the ranks)")
    if args.dry_launch:
        return dry_launch(args, world, rank)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    n_dev = torch.cuda.device_count()
    shared_gpu = world > n_dev           # more ranks than GPUs: only as a functional check (DVAE_ALLOW_SHARED_GPU=1)
    if shared_gpu and os.environ.get("DVAE_ALLOW_SHARED_GPU", "0") != "1":
        raise SystemExit(f"bench.py: {world} ranks but only {n_dev} visible GPU(s): one process per GPU is the design "
                         "(DVAE_ALLOW_SHARED_GPU=1 runs ranks on shared devices as a functional check, not a measurement)")
    local = local % max(1, n_dev)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    force_ddp = os.environ.get("DVAE_FORCE_DDP", "0") == "1"      # exercise the RCCL path with one rank (testing)
    dp = world > 1 or force_ddp
    backend = os.environ.get("DVAE_DIST_BACKEND", "nccl")          # "nccl" IS RCCL on ROCm; gloo only for shared-GPU checks
    if dp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        if os.environ.get("DVAE_RCCL_MAX_CHANNELS"):
            # bound the CUs an in-flight collective occupies beside the W_hh-resident recurrences (which want every CU)
            os.environ["NCCL_MAX_NCHANNELS"] = os.environ["DVAE_RCCL_MAX_CHANNELS"]
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
        if dist.get_world_size() != args.gpus:
            # never a line that claims N GPUs from another number of cooperating ranks
            raise SystemExit(f"bench.py: --gpus {args.gpus} but the process group has {dist.get_world_size()} ranks")

    from dvae_amd import ops
    from dvae_amd.data import SyntheticPairs
    if shared_gpu:
        # two persistent grids cannot both be resident on one GPU (each wants every CU): per-frame recurrences
        ops.LSTM_PERSISTENT = False

    B, T = args.batch, args.frames
    w = build_trainer(dev, B, T, dtype, world, rank, force_ddp)
    data = SyntheticPairs(B, T, n_speakers=10, seed=1234 + rank, device=dev)
    x1, x2, spk = data.batch()
    n_params = sum(p.numel() for p in w.model.parameters())

    cdev = dev if backend == "nccl" else torch.device("cpu")     # where the small control tensors of the collectives live

    def barrier():
        torch.cuda.synchronize()
        if dp:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(n_steps, sync_each=False):
        """EXACTLY n_steps full train steps between two barriers; returns (seconds: max over ranks, last losses, own s)."""
        barrier()
        t0 = time.perf_counter()
        last = None
        for _ in range(n_steps):
            last = w.step(x1, x2, spk, train=True) if sync_each else w.step_async(x1, x2, spk)
        barrier()
        own = time.perf_counter() - t0
        el = own
        if dp:
            tmax = torch.tensor([own], device=cdev, dtype=torch.float64)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            el = float(tmax.item())
        return el, last, own

    extra = {}
    launch = "eager"
    if not dp:
        # ---------------- one GPU: the graph-replayed step is the product path
        use_graph = bool(args.graph)
        if use_graph:
            w.enable_graph(True)
            launch = "hipGraph replay"
        log(f"model built ({n_params} params), {dtype}, warm-up x{args.warmup}")
        for _ in range(max(args.warmup, 2 if use_graph else 0)):   # graph mode: call 1 eager, call 2 captures
            w.step(x1, x2, spk, train=True)
        log("timed region start")
        elapsed, last, _ = timed(args.steps)
        last = tuple(last.tolist())
        ops.lstm_pers_check()
        log(f"timed region done: {1e3 * elapsed / args.steps:.2f} ms/step")
        n_sync = max(1, min(args.steps, 10))
        el_sync, _, _ = timed(n_sync, sync_each=True)
        extra["ms_per_step_sync"] = 1e3 * el_sync / n_sync
        extra["ms_per_step_sync_note"] = (f"{n_sync} steps through step(): + one device->host copy of the 8 loss scalars and the "
                                          "persistent-launch error check per step (the reference syncs 8 times per step)")
    else:
        # ---------------- data parallel: eager first, then the graph with the collectives captured, if it captures
        log(f"rank {rank}/{world}: model built ({n_params} params), {dtype}, EAGER data-parallel warm-up x{args.warmup}")
        import threading

        def first_wedged():
            # the first (default) exchange has no measured predecessor to fall back on: a marked line without a value, and
            # every rank leaves with WATCHDOG_RC (rank 0 first; the others wait so that no launcher takes it down earlier)
            if rank == 0:
                print(json.dumps({"metric": f"utterances/sec (B={B}, 80-mel, T={T}) train step", "value": None,
                                  "unit": "utterances/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                                  "error": "the default data-parallel exchange did not finish its warm-up + timed region "
                                           f"within DVAE_BENCH_FIRST_TIMEOUT; every rank exits {WATCHDOG_RC}",
                                  "variant_watchdog_rc": WATCHDOG_RC}), flush=True)
            else:
                time.sleep(3.0)
            os._exit(WATCHDOG_RC)
        fdog = threading.Timer(float(os.environ.get("DVAE_BENCH_FIRST_TIMEOUT", "420")), first_wedged)
        fdog.daemon = True
        fdog.start()
        for _ in range(max(args.warmup, 1)):
            w.step(x1, x2, spk, train=True)
        el_e, last, own_e = timed(args.steps)
        fdog.cancel()
        last = tuple(last.tolist())
        ms_eager = 1e3 * el_e / args.steps
        per_rank = torch.zeros(world, device=cdev, dtype=torch.float64)
        per_rank[rank] = 1e3 * own_e / args.steps
        dist.all_reduce(per_rank)
        red = w.reducer
        extra.update({"ms_per_step_eager": ms_eager, "per_rank_ms_eager": [round(v, 3) for v in per_rank.tolist()],
                      "rccl_ranks": dist.get_world_size(), "visible_devices": torch.cuda.device_count(),
                      "kfd_gpu_nodes": visible_gpu_count(),
                      "backend": backend, "ranks_share_a_gpu": bool(shared_gpu),
                      "ddp_mode": red.mode, "ddp_issue": red.issue,
                      "rccl_max_channels": os.environ.get("NCCL_MAX_NCHANNELS"),
                      "buckets": {"count": len(red.buckets), "bytes": [4 * (hi - lo) for lo, hi in red.buckets],
                                  "launched_from_backward_hooks": red.stats["hook"], "left_for_finish": red.stats["finish"],
                                  "steps": red.stats["steps"]}})
        log(f"rank {rank}: eager data-parallel step {ms_eager:.2f} ms")
        elapsed, launch = el_e, f"eager ({red.mode}, collectives issued from {red.issue})"
        # No multi-GPU box has ever been available to choose between the exchange variants, so the first one that is measures
        # them all: bucketed all-reduce or reduce-scatter + sharded Adam + all-gather, each with its collectives issued from the
        # backward hooks (overlap) or after backward (nothing beside the W_hh-resident recurrences).  Every variant computes
        # the same update, so the replicas stay in step; each is timed over EXACTLY --steps steps like the first.  The
        # HEADLINE is the default variant (train.py's), measured first; the fastest is reported beside it (`ddp_fastest`), all
        # are listed.  DVAE_BENCH_DDP_VARIANTS="" keeps only the configured one.
        default_variant = f"{red.mode}:{red.issue}"      # what train.py runs (DVAE_DDP_MODE / DVAE_DDP_ISSUE unset): the HEADLINE
        extra["ddp_default"] = {"variant": default_variant, "ms_per_step": ms_eager,
                                "value": world * B * args.steps / el_e,
                                "note": "the exchange train.py uses by default: `value` / `ms_per_step` of this line are this variant's"}
        variants = {default_variant: ms_eager}
        fastest = {"variant": default_variant, "ms_per_step": ms_eager}
        want = os.environ.get("DVAE_BENCH_DDP_VARIANTS", "all_reduce:hook,all_reduce:finish,rs_ag:hook,rs_ag:finish")
        from dvae_amd import ddp as _ddp
        # A variant that WEDGES (a collective some rank never enters) must not cost the line either: from here on a watchdog
        # per rank ends the process once the remaining measurements have taken DVAE_BENCH_VARIANT_TIMEOUT seconds; rank 0
        # first prints the line of what has been measured (a complete timed region of at least the first variant), marked.
        import threading
        wd = {"elapsed": elapsed, "last": last, "launch": launch, "now": "setup"}

        def variants_wedged():
            # A wedged GPU process is NOT rc 0: rank 0 prints the marked line of what was measured (a complete timed region
            # of at least the first variant), then every rank leaves with WATCHDOG_RC.  The other ranks wait a moment
            # first, so that neither this repository's launcher nor torch.distributed.run (both stop the remaining ranks
            # at the first non-zero exit) can take rank 0 down before its line is out.
            if rank == 0:
                ex = dict(extra, ddp_variants_ms_per_step=dict(variants), variant_watchdog_rc=WATCHDOG_RC,
                          variant_watchdog=f"fired during {wd['now']}: the line is the default variant, measured before; "
                                           f"every rank exits {WATCHDOG_RC}")
                print(json.dumps(_result(args, world, B, T, dtype, n_params, wd["elapsed"], wd["last"], wd["launch"], ex, None)),
                      flush=True)
            else:
                time.sleep(3.0)
            os._exit(WATCHDOG_RC)
        vdog = threading.Timer(float(os.environ.get("DVAE_BENCH_VARIANT_TIMEOUT", "300")), variants_wedged)
        vdog.daemon = True
        vdog.start()
        for v in [x for x in want.split(",") if x and x not in variants]:
            wd["now"] = v
            try:
                mode_v, issue_v = v.split(":")
                r2 = red if mode_v == red.mode else _ddp.GradReducer(w.optimizer.flat_g, w.optimizer.names, w.optimizer.params,
                                                                       w.optimizer.offsets, mode=mode_v, issue=issue_v)
                r2.issue, r2.force = issue_v, red.force
                w.attach_reducer(r2)
                for _ in range(2):
                    w.step_async(x1, x2, spk)
                el_v, last_v, _ = timed(args.steps)
                # a sharded step leaves every rank with current Adam moments for its own slices only: collect them, or the
                # full-Adam variants and measurements that follow would run on replicas that drift apart
                r2.gather_moments(w.optimizer)
                variants[v] = 1e3 * el_v / args.steps
                log(f"rank {rank}: {v}: {variants[v]:.2f} ms")
                if variants[v] < fastest["ms_per_step"]:      # reported beside the headline, never instead of it
                    fastest = {"variant": v, "ms_per_step": variants[v]}
            except Exception as e:      # a variant that fails must not cost the line
                variants[v] = "failed: " + repr(e)[:160]
        wd["now"] = "the measurements after the variants"
        extra["ddp_variants_ms_per_step"] = variants
        fastest["value"] = world * B / (fastest["ms_per_step"] * 1e-3)
        extra["ddp_fastest"] = fastest
        red.issue = os.environ.get("DVAE_DDP_ISSUE", "finish")
        w.attach_reducer(red)            # (sets fold_zero_grad for the reducer's mode, both directions)
        # graph attempt, guarded three ways: try/except around the capture, agreement of all ranks, and a watchdog that
        # prints the eager-only line and ends the process if the attempt hangs
        # OPT-IN (DVAE_BENCH_DDP_GRAPH=1): no multi-rank run has shown the captured step equal to the eager one, so it is
        # neither attempted nor allowed to become the headline by default
        if args.graph and os.environ.get("DVAE_BENCH_DDP_GRAPH", "0") == "1":
            import threading
            state = {"line": None}

            def give_up():
                if rank == 0 and state["line"] is not None:
                    print(state["line"], flush=True)
                os._exit(3)             # a wedged attempt is a failure, even though the eager line above is valid
            state["line"] = json.dumps(_result(args, world, B, T, dtype, n_params, elapsed, last, launch,
                                               dict(extra, graph_error="graph attempt exceeded its time limit"), None))
            dog = threading.Timer(float(os.environ.get("DVAE_BENCH_GRAPH_TIMEOUT", "150")), give_up)
            dog.daemon = True
            dog.start()
            ok, err = 1, None
            try:
                w.enable_graph(True, ddp=True)
                for _ in range(3):              # call 1 eager, call 2 captures + replays, call 3 replays
                    w.step(x1, x2, spk, train=True)
                if w.graph_fallback is not None:
                    ok, err = 0, w.graph_fallback
            except Exception as e:
                ok, err = 0, repr(e)[:300]
            flag = torch.tensor([ok], device=cdev, dtype=torch.int32)
            try:
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                ok = int(flag.item())
            except Exception as e:
                ok, err = 0, (err or "") + " | agreement all-reduce failed: " + repr(e)[:200]
            if ok:
                el_g, last_g, _ = timed(args.steps)
                ms_graph = 1e3 * el_g / args.steps
                extra["ms_per_step_graph"] = ms_graph
                if ms_graph < ms_eager:
                    elapsed, launch, last = el_g, "hipGraph replay (RCCL all-reduce captured in the graph)", tuple(last_g.tolist())
            else:
                w.enable_graph(False)
                extra["graph_error"] = err or "another rank failed to capture"
            dog.cancel()
        # exposed communication, measured LAST (the replicas drift apart without the exchange): the same eager step with
        # the reducer detached, i.e. no collective at all
        w.enable_graph(False)
        w.attach_reducer(None)
        for _ in range(2):
            w.step_async(x1, x2, spk)
        n_nr = max(1, min(args.steps, 10))
        el_nr, _, _ = timed(n_nr)
        w.attach_reducer(red)
        extra["ms_per_step_no_allreduce"] = 1e3 * el_nr / n_nr
        extra["allreduce_exposed_ms"] = ms_eager - 1e3 * el_nr / n_nr
        extra["allreduce_exposed_variant"] = next(iter(variants))          # the first variant measured (ms_eager)
        # the single-rank product step (hipGraph replay, no reducer) in THIS process, every rank at once on its own GPU: what
        # `value` is to be held against (the driver computes the efficiency itself from its own N = 1 run; DVAE_BENCH_N1_MS
        # supplies a number measured elsewhere instead)
        n1_ms, n1_src = None, None
        if os.environ.get("DVAE_BENCH_N1_MS"):
            n1_ms, n1_src = float(os.environ["DVAE_BENCH_N1_MS"]), "DVAE_BENCH_N1_MS (supplied)"
        elif args.graph:
            # the capture may fail on ONE rank: the ranks agree on it (an all-reduce outside the try) BEFORE any of them
            # enters timed(), whose barriers would otherwise hold the others until the watchdog ends the whole job
            ok1, err1 = 1, None
            try:
                w.attach_reducer(None)
                w.enable_graph(True)
                for _ in range(3):
                    w.step_async(x1, x2, spk)
                torch.cuda.synchronize()
            except Exception as e:
                ok1, err1 = 0, repr(e)[:200]
            flag1 = torch.tensor([ok1], device=cdev, dtype=torch.int32)
            dist.all_reduce(flag1, op=dist.ReduceOp.MIN)
            if int(flag1.item()):
                el_1, _, _ = timed(n_nr)
                n1_ms = 1e3 * el_1 / n_nr
                n1_src = (f"in-process: {n_nr} graph-replayed steps with the reducer detached, all {world} ranks at once "
                          "(max over ranks)")
            else:
                extra["scaling_vs_n1_error"] = err1 or "another rank failed to capture the single-rank step"
            w.enable_graph(False)
            w.attach_reducer(red)
        if n1_ms:
            n1_value = B / (n1_ms * 1e-3)
            extra["scaling_vs_n1"] = {"n1_ms_per_step": n1_ms, "n1_value": n1_value, "source": n1_src,
                                      "speedup": (world * B * args.steps / elapsed) / n1_value, "ideal": world}
        ops.lstm_pers_check()
        vdog.cancel()

    roof = roof_lstm = None
    if dp:
        w.attach_reducer(None)           # the roofline pass below runs on rank 0 alone: no collective in it
    if not args.no_roofline and rank == 0:
        # the SAME kernels, timed with HIP events around every launch over eager steps right after the timed region
        # (N > 1: rank 0's, with the reducer detached; the other ranks wait in the final barrier)
        prof_steps = min(args.steps, 3)
        tags, (ms, launches, flops), tags2, tot2 = profile_families(w, x1, x2, spk, ops, prof_steps)
        roof_lstm = lstm_roofline(tags2, tot2, prof_steps)
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
            roof = {"bound": "mfma", "kernel": "contraction family: " + ("gemm_bf16_256_kernel / gemm_bf16_tall_kernel / gemm_f32_kernel<MODE=1>" if dtype == "bf16" else "gemm_f32_kernel<...> + gemm_x3_tall_kernel<...>") + " (" + ARITH[dtype] + ")",
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
                                        "tflops": t["flops"] / (t["ms"] * 1e-3) / 1e12,
                                        "algorithmic_bytes_per_launch": t["bytes"] / max(1, t["launches"])} for t in tags],
                    "timed": f"HIP events around every launch, {prof_steps} eager steps right after the graph-replayed "
                             "timed region"}

    final_line = None
    if rank == 0:
        out = _result(args, world, B, T, dtype, n_params, elapsed, last, launch, extra, roof)
        if roof_lstm:
            out["roofline_lstm"] = roof_lstm
        del w
        torch.cuda.empty_cache()
        if not dp and not args.no_other_configs and (B, T) == (64, 128):
            others = []
            for name, b, t in (("configs[2]: 1xMI355X bf16, B=128, T=256", 128, 256),
                               ("configs[4] per-GPU shape: bf16, B=64, T=512", 64, 512)):
                try:
                    log(f"timing {name}")
                    others.append(time_other_config(dev, name, b, t, "bf16"))
                except Exception as e:      # never lose the headline line to a side measurement
                    others.append({"config": name, "error": repr(e)[:300]})
            out["other_configs"] = others
    if dp:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        if not args.no_cpu_baseline:
            # (N > 1: after the process group is gone — the other ranks have left, the host cores are free)
            log("timing the CPU oracle on the host cores (child processes, bounded)")
            cb = cpu_baseline(B, T)
            out["cpu_baseline"] = cb
            if cb.get("value"):
                out["speedup_vs_cpu_baseline"] = out["value"] / cb["value"]
        final_line = json.dumps(out)
        # RCCL writes a banner through C stdio, which sits in libc's buffer until exit: flush it first so that the
        # JSON line is the LAST line of stdout
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(final_line, flush=True)


def _result(args, world, B, T, dtype, n_params, elapsed, last, launch, extra, roof):
    ms_step = 1e3 * elapsed / args.steps
    value = world * B * args.steps / elapsed
    step_flops = algorithmic_flops_per_pair(T) * B
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
                      "global_batch": world * B, "frames": T, "parallelism": f"dp{world}", "launch": launch,
                      "params": n_params},
           "step_tflops_algorithmic": step_flops / 1e12,
           "step_frac_of_peak": step_flops / (ms_step * 1e-3) / 1e12 / PEAKS[dtype],
           "final_loss": last[0] if last else None}
    out.update(extra)
    if roof:
        out["roofline"] = roof
    return out


if __name__ == "__main__":
    main()
