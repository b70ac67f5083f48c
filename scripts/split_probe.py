"""GPU box: one contraction shape against the number of k-splits (atomic epilogue into a zeroed result; the zero-fill is
timed with it).  usage: split_probe.py M N K a_kc b_kc [splits...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dvae_amd  # noqa: F401
from dvae_amd import ops
from dvae_amd._lib import check, lib, ptr, stream

M, N, K, akc, bkc = (int(v) for v in sys.argv[1:6])
splits = [int(v) for v in sys.argv[6:]] or [1, 2, 4, 8]
t = lambda *s: torch.randn(*s, device="cuda")
A = t(M, K) if akc else t(K, M)
B = t(N, K) if bkc else t(K, N)
Cc = torch.empty(M, N, device="cuda")
lda, ldb = (K if akc else M), (K if bkc else N)
L = lib()
for sk in splits:
    def fn():
        if sk == 1:
            ops.gemm(A, B, Cc, None, M, N, K, lda, ldb, N, bool(akc), bool(bkc), 0, ops.EPI_STORE, 1)
        else:
            check(L.dvae_zero_f32(ptr(Cc), Cc.numel(), stream()), "zero")
            ops.gemm(A, B, Cc, None, M, N, K, lda, ldb, N, bool(akc), bool(bkc), 0, ops.EPI_ATOMIC, sk)
    for _ in range(30):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(300):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 300
    print(f"M={M} N={N} K={K} akc={akc} bkc={bkc} sk={sk}: {ms * 1e3:.1f} us  {2.0 * M * N * K / ms / 1e9:.1f} TF/s", flush=True)
