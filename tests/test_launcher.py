"""`--gpus N` means N ranks (missing 1): bench.py started without a rendezvous in the environment launches N
fresh rank processes before anything touches the GPU; started under torch.distributed.run it refuses a WORLD_SIZE that
differs from --gpus.  Exercised here on CPU with --dry-launch (gloo, no GPU work): the launcher, the rendezvous, the
barrier and the max-over-ranks timing are the real ones."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(OMP_NUM_THREADS="1", **kw)
    return env


def test_gpus_2_launches_two_cooperating_ranks():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "3", "--warmup", "0", "--dry-launch"],
                       env=_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    out = json.loads(lines[-1])                       # the JSON line is the LAST line of the launcher's stdout
    assert out["n_gpus"] == 2 and out["dry_launch"] is True
    assert out["config"]["parallelism"] == "dp2"
    # the slowest rank (rank 1 sleeps 2 ms per step) sets the time: max over ranks, not rank 0's own
    assert out["ms_per_step"] >= 1.9, out
    assert sum(l.lstrip().startswith("{") for l in lines) == 1          # ONE JSON line
    # the keys every N > 1 line carries: the ranks that really cooperated, the DEFAULT exchange — the
    # one train.py runs, which is what `value` is — and the fastest variant beside it, never instead of it
    assert out["rccl_ranks"] == 2
    assert set(out["ddp_default"]) >= {"variant", "ms_per_step", "value"} and out["ddp_default"]["variant"] == "all_reduce:finish"
    assert set(out["ddp_fastest"]) >= {"variant", "ms_per_step", "value"}
    assert "ddp_variants_ms_per_step" in out


def test_world_size_mismatch_is_an_error_not_a_silent_single_rank_run():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "8", "--dry-launch"],
                       env=_env(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"), capture_output=True, text=True, timeout=120)
    assert r.returncode != 0
    assert "WORLD_SIZE=1" in r.stderr and "--gpus 8" in r.stderr
    assert not r.stdout.strip()


def test_launcher_propagates_a_failing_rank():
    # rank processes of a REAL (non-dry) run fail here: no GPU.  The launcher must come back non-zero, not hang
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=_env(DVAE_ALLOW_SHARED_GPU="1"), capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert not any(l.lstrip().startswith("{") for l in r.stdout.splitlines())


def test_train_cli_has_the_data_parallel_flags():
    sys.path.insert(0, ROOT)
    import dvae_amd  # noqa: F401
    from dvae_amd import train
    a = train.get_parse().parse_args(["--gpus", "4", "--batch-size", "8"])
    assert a.gpus == 4 and a.gpu_loader == -1
    assert callable(train.launch_ranks) and callable(train.setup_data_parallel)


def test_launcher_counts_gpus_without_the_hip_runtime(tmp_path, monkeypatch):
    """The launcher process never touches the HIP runtime: the device count comes from the KFD topology in sysfs (nodes with
    SIMDs), cut down by the *_VISIBLE_DEVICES lists; None (then --gpus is trusted) where there is no topology."""
    sys.path.insert(0, ROOT)
    import bench
    src = open(BENCH).read()
    body = src[src.index("def launch_ranks("):src.index("def dry_launch(")]
    assert "torch.cuda" not in body                      # nothing of the runtime in the launcher
    import glob as _glob
    nodes = tmp_path / "nodes"
    for i, simd in enumerate((0, 0, 256, 256, 256)):   # two CPU nodes, three GPUs
        d = nodes / str(i)
        d.mkdir(parents=True)
        (d / "properties").write_text(f"cpu_cores_count {64 if simd == 0 else 0}\nsimd_count {simd}\nmem_banks_count 1\n")
    real = _glob.glob
    monkeypatch.setattr(_glob, "glob", lambda pat: real(str(nodes / "*" / "properties")) if "kfd" in pat else real(pat))
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        monkeypatch.delenv(var, raising=False)
    assert bench.visible_gpu_count() == 3
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0,2")
    assert bench.visible_gpu_count() == 2
    monkeypatch.setattr(_glob, "glob", lambda pat: [] if "kfd" in pat else real(pat))
    assert bench.visible_gpu_count() is None
