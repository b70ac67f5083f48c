"""GPU box: the M = 128 (2B segments) x huge-weight contractions (enc_linear / dec_pre_linear2 forward and data gradient):
time against the split-K factor (atomic epilogue traffic = splits x 1 MB)."""
import os as _os
_os.environ.setdefault("DVAE_LIB_PATH", _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
                                                       "disentangle-vae-for-vc_amd", "libdvae_dev.so"))
import sys
sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
import torch
import dvae_amd  # noqa
from dvae_amd import ops

dev = "cuda"
for name, M, N, K, bkc, sks in (("fwd [128 x 2048 x 16384]", 128, 2048, 16384, 1, (4, 8, 16, 32, 64)),
                                ("dgrad [128 x 16384 x 2048]", 128, 16384, 2048, 0, (1, 2, 4, 8)),
                                ("fwd2 [128 x 16384 x 2048] NT", 128, 16384, 2048, 1, (1, 2, 4, 8)),
                                ("dgrad2 [128 x 2048 x 16384] NN", 128, 2048, 16384, 0, (4, 8, 16, 32, 64))):
    A = torch.randn(M, K, device=dev)
    B = torch.randn((N, K) if bkc else (K, N), device=dev)
    Cc = torch.zeros(M, N, device=dev)
    for sk in sks:
        epi = 2 if sk > 1 else 0
        fn = lambda: ops.gemm(A, B, Cc, None, M, N, K, K, (K if bkc else N), N, True, bool(bkc), 0, epi, sk, "fp32x3")
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(100):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = 10.0 * e0.elapsed_time(e1)
        print(f"{name:34s} sk={sk:3d} narrow={_os.environ.get('DVAE_GEMM_NARROW', '-')}: {us:7.1f} us  {4.0 * N * K / us / 1e6:5.2f} TB/s of weights")
