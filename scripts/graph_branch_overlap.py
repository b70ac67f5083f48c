"""GPU-box experiment: does a hipGraph run two independent branches concurrently?  Branch A: the H=64 bidirectional
LSTM recurrence (16 workgroups busy, ~0.25 ms per layer-pass); branch B: a weight-gradient GEMM (~1.2 ms, whole chip)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dvae_amd  # noqa
from dvae_amd import ops
T, N, H, In = 128, 128, 64, 512
P = lambda *s: torch.randn(*s, device="cuda") * 0.05
ps = [P(4 * H, In), P(4 * H, H), P(4 * H), P(4 * H), P(4 * H, In), P(4 * H, H), P(4 * H), P(4 * H)]
x = torch.randn(T * N, In, device="cuda")
dg, h, gw = torch.randn(T * N, 4096, device="cuda"), torch.randn(T * N, 1024, device="cuda"), torch.zeros(4096, 1024, device="cuda")
side = torch.cuda.Stream()
def lstm():
    with torch.no_grad():
        for _ in range(4):
            ops.LstmLayerFn.apply(x, T, N, *ps)
def gemm():
    for _ in range(2):
        ops.linear_wgrad_acc(dg, h, gw)
def seq():
    lstm(); gemm()
def par():
    main = torch.cuda.current_stream()
    side.wait_stream(main)
    with torch.cuda.stream(side):
        gemm()
    lstm()
    main.wait_stream(side)
for name, fn in (("lstm only", lstm), ("gemm only", gemm), ("sequential", seq), ("two branches", par)):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    for _ in range(3): g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20): g.replay()
    torch.cuda.synchronize()
    print(f"{name:14s} {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms")
