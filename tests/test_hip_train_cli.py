"""The training CLI (`python -m dvae_amd.train`, mirror of the reference's train.py:13-58,89-99) data parallel on the HIP
kernels: (a) one rank through the real RCCL backend (DVAE_FORCE_DDP=1: collectives issued although world == 1);
(b) `--gpus 2`: the CLI launches two rank processes itself — they share the box's one GPU and exchange through gloo
(DVAE_ALLOW_SHARED_GPU / DVAE_DIST_BACKEND: functional check) — each feeds from its shard of the device-resident corpus,
rank 0 writes the checkpoint."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _corpus(tmp_path):
    sys.path.insert(0, ROOT)
    import dvae_amd  # noqa: F401
    from dvae_amd.data import write_synthetic_corpus
    return write_synthetic_corpus(str(tmp_path / "corpus"), n_speakers=2, n_utt=16, length=96, seed=0)   # 16 pairs


def _cli(tmp_path, extra, env_extra):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(PYTHONPATH=ROOT + os.pathsep + env.get("PYTHONPATH", ""), MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY="0", **env_extra)
    log_dir = tmp_path / "results"
    cmd = [sys.executable, "-c", "import dvae_amd.train as t, sys; t.main(sys.argv[1:])", "--train", "true",
           f"--dataset_fp={_corpus(tmp_path)}", "--batch-size=4", "--latent-size=32", "--speaker_size=4", "--lr=1e-4",
           "--epochs=2", "--report-interval=2", "--mse_cof=10", "--kl_cof=10", f"--log_dir={log_dir}", "--seed=3",
           "--do-not-resume"] + extra
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    recs = [json.loads(l) for l in open(log_dir / "logs" / "DisentangledVAE_VCTK" / "scalars.jsonl")]
    return log_dir, recs, r


def test_train_cli_single_rank_through_rccl(tmp_path):
    log_dir, recs, r = _cli(tmp_path, [], dict(DVAE_FORCE_DDP="1", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1"))
    assert [x["epoch"] for x in recs] == [1, 2]
    assert all(v == v and abs(v) < 1e9 for x in recs for v in x.values())
    assert recs[1]["Loss/Reconstruction Loss1"] < recs[0]["Loss/Reconstruction Loss1"]
    assert (log_dir / "checkpoints" / "DisentangledVAE_VCTK_2.pth").exists()
    assert (log_dir / "checkpoints" / "DisentangledVAE_VCTK_2.opt").exists()


def test_train_cli_gpus_2_launches_two_sharded_ranks(tmp_path):
    log_dir, recs, r = _cli(tmp_path, ["--gpus", "2"], dict(DVAE_ALLOW_SHARED_GPU="1", DVAE_DIST_BACKEND="gloo"))
    assert [x["epoch"] for x in recs] == [1, 2]                    # ONE writer (rank 0), two epochs
    assert recs[1]["Loss/Reconstruction Loss1"] < recs[0]["Loss/Reconstruction Loss1"]
    ck = sorted(p.name for p in (log_dir / "checkpoints").iterdir())
    assert ck == ["DisentangledVAE_VCTK_2.opt", "DisentangledVAE_VCTK_2.pth"], ck
    import torch
    osd = torch.load(log_dir / "checkpoints" / "DisentangledVAE_VCTK_2.opt", map_location="cpu")
    # 16 pairs over 2 ranks x batch 4 = 2 steps per epoch per rank, 2 epochs
    assert osd["t"] == 4, osd["t"]


def test_train_cli_a_failing_rank_ends_the_job_instead_of_hanging_it(tmp_path):
    """a rank that raises must not walk into the closing barrier while its peer sits in an all-reduce — it tears
    its communicators down and leaves with the exception; the launcher sees the non-zero exit and stops the other rank.
    The whole job must be over in seconds, not after a collective's timeout, and the traceback must be there."""
    import time
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(PYTHONPATH=ROOT + os.pathsep + env.get("PYTHONPATH", ""), MASTER_ADDR="127.0.0.1",
               HSA_ENABLE_IPC_MODE_LEGACY="0", DVAE_ALLOW_SHARED_GPU="1", DVAE_DIST_BACKEND="gloo", DVAE_TEST_MODE="1", DVAE_TEST_FAIL_RANK="1")
    cmd = [sys.executable, "-c", "import dvae_amd.train as t, sys; t.main(sys.argv[1:])", "--train", "true",
           f"--dataset_fp={_corpus(tmp_path)}", "--batch-size=4", "--latent-size=32", "--speaker_size=4", "--lr=1e-4",
           "--epochs=50", "--report-interval=50", "--mse_cof=10", "--kl_cof=10", f"--log_dir={tmp_path / 'results'}", "--seed=3",
           "--do-not-resume", "--gpus", "2"]
    t0 = time.time()
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    dt = time.time() - t0
    assert r.returncode != 0, r.stdout[-500:]
    assert "injected failure on rank 1" in r.stderr, r.stderr[-2000:]
    assert dt < 90, f"the job took {dt:.0f} s to end after a rank failed"
    assert not (tmp_path / "results" / "checkpoints" / "DisentangledVAE_VCTK_50.pth").exists()      # rank 0 did not train on alone


def _bench_two_ranks(env_extra, frames="64", batch="4", rc=0):
    """`bench.py --gpus 2` as the driver starts it for N = 1 (no rendezvous in the environment: bench.py launches the two
    ranks itself); the box has one GPU, so the ranks share it and exchange through gloo — a functional check of the N > 1
    line, not a measurement."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", DVAE_ALLOW_SHARED_GPU="1", DVAE_DIST_BACKEND="gloo",
               **env_extra)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--batch", batch, "--frames", frames], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == rc, (r.returncode, r.stdout[-1500:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                       # ONE JSON line, from rank 0
    return json.loads(lines[0])


def test_bench_gpus_2_line_lists_every_exchange_variant():
    out = _bench_two_ranks({})
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["value"] > 0
    v = out["ddp_variants_ms_per_step"]
    assert set(v) == {"all_reduce:hook", "all_reduce:finish", "rs_ag:hook", "rs_ag:finish"}, v
    assert all(isinstance(x, float) and x > 0 for x in v.values()), v
    # round 6: the headline is the DEFAULT exchange (train.py's, measured first), the fastest is reported beside it
    d, f = out["ddp_default"], out["ddp_fastest"]
    assert d["variant"] == "all_reduce:finish" and abs(out["ms_per_step"] - d["ms_per_step"]) < 1e-6 * out["ms_per_step"]
    assert abs(d["ms_per_step"] - v[d["variant"]]) < 1e-9 and abs(out["value"] - d["value"]) < 1e-6 * out["value"]
    assert abs(f["ms_per_step"] - min(v.values())) < 1e-9 and v[f["variant"]] == f["ms_per_step"] and f["value"] >= d["value"] * 0.999
    assert out["rccl_ranks"] == 2
    assert "variant_watchdog" not in out
    # the N > 1 line carries everything the N = 1 line does: the contraction family's roofline (rank 0,
    # reducer detached), the recurrence family's, the CPU baseline (after the process group is gone), and the single-rank
    # step of the same process to hold `value` against
    rl = out["roofline"]
    assert rl["bound"] == "mfma" and rl["achieved"] > 0 and 0 < rl["frac"] < 1 and rl["launches_per_step"] > 10
    assert out["roofline_lstm"]["kernel_ms_per_step"] > 0
    cb = out["cpu_baseline"]
    assert cb["kind"] == "port" and cb["value"] > 0 and cb["cores"] >= 1
    sv = out["scaling_vs_n1"]
    assert sv["n1_ms_per_step"] > 0 and sv["ideal"] == 2 and 0 < sv["speedup"] < 2.5, sv
    n1_keys = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
               "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"}
    assert n1_keys <= set(out), n1_keys - set(out)


def test_bench_gpus_2_a_wedged_variant_does_not_cost_the_line():
    """DVAE_BENCH_VARIANT_TIMEOUT = 0.05 s: the watchdog fires while the second exchange variant is being set up or timed;
    rank 0 must still print the (marked) line of the first variant's complete timed region — and every rank must then
    leave with the watchdog's exit code, which the launcher hands on: a wedged GPU process is not rc 0."""
    out = _bench_two_ranks({"DVAE_BENCH_VARIANT_TIMEOUT": "0.05"}, rc=5)
    assert out["n_gpus"] == 2 and out["value"] > 0
    assert "fired during" in out["variant_watchdog"] and out["variant_watchdog_rc"] == 5
    assert "all_reduce:finish" in out["ddp_variants_ms_per_step"]      # the first one measured
