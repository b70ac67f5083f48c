"""GPU box: do the k-split products with few rows (M = 128; atomically accumulated C with a row stride of 8 / 64 KB) suffer
from the power-of-two row stride of C?  The same products into a C whose rows are padded by 32 / 64 / 160 floats."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dvae_amd  # noqa
from dvae_amd import ops
for (M, N, K, sk) in [(128, 16384, 2048, 4), (128, 2048, 16384, 32), (2048, 128, 16384, 32), (80, 1024, 16384, 64)]:
    A = torch.randn(M, K, device="cuda")
    B = torch.randn(N, K, device="cuda")
    for pad in (0, 32, 64, 160):
        ldc = N + pad
        Cc = torch.zeros(M, ldc, device="cuda")
        fn = lambda: ops.gemm(A, B, Cc, None, M, N, K, K, K, ldc, True, True, 0, 2, sk, "fp32x3")
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(300):
            fn()
        e1.record()
        torch.cuda.synchronize()
        print(f"M={M} N={N} K={K} sk={sk} ldc=N+{pad}: {1e3 * e0.elapsed_time(e1) / 300:.1f} us")
