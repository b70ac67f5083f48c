"""GPU-box: what the weight-gradient contractions would cost with BOTH operands k-contiguous (transposed activations):
the same M x N x K, split-K and atomic epilogue through the T,T variant.  usage: wgrad_tt_potential.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dvae_amd  # noqa
from dvae_amd import ops
def timeit(fn, fl, name):
    for _ in range(30): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 100
    print(f"{name:44s} {ms * 1e3:8.1f} us {fl / ms / 1e9:7.1f} TF/s")
R = 16384
for M, N, sk, nm in ((512, 2560, 6, "conv 512->512 (5 taps side by side)"), (4096, 1024, 2, "W_hh / W_ih H=1024"), (4096, 512, 4, "W_ih dec_lstm2.0")):
    a_t, b_t = torch.randn(M, R, device="cuda"), torch.randn(N, R, device="cuda")      # k-contiguous
    a_r, b_r = torch.randn(R, M, device="cuda"), torch.randn(R, N, device="cuda")      # row-contiguous (today)
    c = torch.zeros(M, N, device="cuda")
    fl = 2.0 * M * N * R
    timeit(lambda: ops.gemm(a_r, b_r, c, None, M, N, R, M, N, N, False, False, 0, ops.EPI_ATOMIC, sk), fl, nm + " F,F")
    timeit(lambda: ops.gemm(a_t, b_t, c, None, M, N, R, R, R, N, True, True, 0, ops.EPI_ATOMIC, sk), fl, nm + " T,T")
    timeit(lambda: ops.gemm(a_t, b_r, c, None, M, N, R, R, N, N, True, False, 0, ops.EPI_ATOMIC, sk), fl, nm + " T,F")
