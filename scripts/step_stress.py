"""GPU box: is the whole train step independent of what ELSE the GPU is doing?  Deterministic mode (ops.set_deterministic:
two runs are bit-identical), B = 64, T = 128 (configs[1]: the persistent recurrences, k-split backward, tall contraction
kernels as the benchmark runs them), graph replay, five inputs cycled, lr > 0.  Run 1: alone.  Run 2: the same steps while a
second stream streams GiBs through HBM (what an overlapped RCCL all-reduce does to the memory system).  Every loss of
every step and the final parameters / moments must agree BIT FOR BIT.
usage: step_stress.py [steps=60] [B=64] [T=128]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from dvae_amd import ops
from dvae_amd.data import SyntheticPairs

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
T = int(sys.argv[3]) if len(sys.argv) > 3 else 128
dev = torch.device("cuda", 0)
ops.set_deterministic(os.environ.get("DVAE_DETERMINISTIC", "0") == "1")      # round 6: the DEFAULT mode is bit-identical too (run with and without)
data = [SyntheticPairs(B, T, n_speakers=10, seed=100 + i, device=dev).batch() for i in range(5)]
gen = torch.Generator(device="cpu").manual_seed(0)
noise = [(torch.randn(B, 28, generator=gen), torch.randn(B, 28, generator=gen), torch.randn(B, 4, generator=gen)) for _ in range(5)]


def run(load):
    w = bench.build_trainer(dev, B, T, os.environ.get("STRESS_DTYPE", "fp32x3"))      # STRESS_DTYPE=bf16: the bf16 mode (its recurrences hand over inside one XCD, round 6)
    w.enable_graph(True)
    side = torch.cuda.Stream()
    a = torch.empty(1 << 28, device=dev, dtype=torch.float32)
    b = torch.empty_like(a)
    out = []
    for i in range(steps):
        if load:
            with torch.cuda.stream(side):
                for _ in range(1 + i % 3):
                    b.copy_(a)
        x1, x2, spk = data[i % 5]
        w.model.eps_override = noise[(3 * i) % 5]
        out.append(w.step(x1, x2, spk, train=True))
    torch.cuda.synchronize()
    return out, w.optimizer.flat_p.clone(), w.optimizer.exp_avg.clone(), w.optimizer.exp_avg_sq.clone()


ref = run(False)
again = run(False)
got = run(True)
for name, x in (("alone, second run", again), ("under foreign HBM traffic", got)):
    bad = [i for i in range(steps) if x[0][i] != ref[0][i]]
    same = all(torch.equal(p, q) for p, q in zip(x[1:], ref[1:]))
    print(f"{name}: {len(bad)} of {steps} steps differ in a loss" + (f" (first: step {bad[0]}: {x[0][bad[0]][0]} vs {ref[0][bad[0]][0]})" if bad else "")
          + f"; parameters and moments bit-identical: {same}")
print("final loss", ref[0][-1][0], "first", ref[0][0][0])
