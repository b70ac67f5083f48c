"""GPU box: the short / narrow contraction shapes of the B=64, T=128 step (fp32x3), one by one: us and TF/s.
usage: small_shapes.py [reps]   — dispatch knobs (DVAE_GEMM_TALL / NARROW / BK / TALL_MIN) act in the dev build only."""
import os as _os
_os.environ.setdefault("DVAE_LIB_PATH", _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
                                                       "disentangle-vae-for-vc_amd", "libdvae_dev.so"))
import sys
sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
import torch
import dvae_amd  # noqa
from dvae_amd import ops

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = "cuda"
SHAPES = [  # name, M, N, K, a_kc, b_kc, epi, sk
    ("P1 proj K=128", 16384, 2048, 128, 1, 1, 0, 1), ("P2 dx N=128", 16384, 128, 2048, 1, 1, 0, 1),
    ("P3 proj N=256 K=512", 16384, 256, 512, 1, 1, 0, 1), ("P4 N=512 K=256", 16384, 512, 256, 1, 1, 0, 1),
    ("P5 N=256 K=128", 16384, 256, 128, 1, 1, 0, 1), ("P6 N=128 K=256", 16384, 128, 256, 1, 1, 0, 1),
    ("P7 M=128 N=16384 K=2048", 128, 16384, 2048, 1, 1, 2, 4), ("P8 M=128 N=2048 K=16384", 128, 2048, 16384, 1, 1, 2, 32),
    ("W1 outer 2048x16384 K=128 store", 2048, 16384, 128, 0, 0, 0, 1), ("W2 outer 16384x2048 K=128 store", 16384, 2048, 128, 0, 0, 0, 1),
    ("W3 256x64 K=16256", 256, 64, 16256, 0, 0, 2, 63), ("W4 256x512 K=16384", 256, 512, 16384, 0, 0, 2, 64),
    ("W5 256x128 K=16384", 256, 128, 16384, 0, 0, 2, 64), ("W6 2048x128 K=16384", 2048, 128, 16384, 0, 0, 2, 32),
    ("W7 80x1024 K=16384", 80, 1024, 16384, 0, 0, 2, 64), ("P9 N=80 K=1024", 16384, 80, 1024, 1, 1, 2, 4),
    ("P10 N=1024 K=80", 16384, 1024, 80, 1, 0, 0, 1),
]
tot = 0.0
for name, M, N, K, akc, bkc, epi, sk in SHAPES:
    A = torch.randn((M, K) if akc else (K, M), device=dev)
    B = torch.randn((N, K) if bkc else (K, N), device=dev)
    Cc = torch.zeros(M, N, device=dev)
    lda, ldb = (K if akc else M), (K if bkc else N)
    fn = lambda: ops.gemm(A, B, Cc, None, M, N, K, lda, ldb, N, bool(akc), bool(bkc), 0, epi, sk, "fp32x3")
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = 1e3 * e0.elapsed_time(e1) / reps
    tot += us
    by = 4.0 * (M * K + N * K + M * N * (2 if epi else 1))
    print(f"{us:8.1f} us {2.0 * M * N * K / us / 1e6:7.1f} TF/s {by / us / 1e6:6.2f} TB/s(alg)  {name}")
print(f"sum {tot:.0f} us")
