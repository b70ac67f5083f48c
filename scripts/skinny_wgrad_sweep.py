"""GPU box: the H = 64 LSTM weight gradients (dW[4H=256, In] = dG^T x over K = T*N rows, atomically accumulated split-k)
against the number of splits, in the default arithmetic (K = 16384) and the bf16 mode (K = 65536, fp32 operands)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dvae_amd import ops


def timeit(fn, reps=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for dtype, K in (("fp32x3", 16384), ("bf16", 65536)):
    with ops.compute_dtype(dtype):
        mode = ops.current_mode()
        for M, N in ((256, 64), (256, 128), (256, 512)):
            dg = torch.randn(K, M, device="cuda")
            x = torch.randn(K, N, device="cuda")
            dw = torch.zeros(M, N, device="cuda")
            auto = ops._split_k(ops._tiles(M, N), K)
            row = []
            for sk in sorted({4, 8, 16, 32, 64, 128, 255, auto}):
                if sk > K // 256:
                    continue
                us = timeit(lambda: ops.gemm(dg, x, dw, None, M, N, K, M, N, N, False, False, ops.ACT_NONE, ops.EPI_ATOMIC, sk, mode))
                row.append(f"{sk}{'*' if sk == auto else ''}:{us:.0f}")
            print(f"{dtype} dW[{M}x{N}] K={K}  us by splits (* = ops._split_k):", " ".join(row), flush=True)
