"""GPU box: the conv weight gradient (five per-tap outputs, split over the rows with atomic epilogues) against the number
of k-splits, at BASELINE configs[2]'s bf16 shape (R = 65536 rows) and the default arithmetic's (R = 16384)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dvae_amd import ops
from dvae_amd._lib import check, lib, ptr, stream

L = lib()


def timeit(fn, reps=20):
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


for dtype, R, nseg in (("bf16", 65536, 256), ("bf16", 32768, 64), ("fp32x3", 16384, 128)):
    with ops.compute_dtype(dtype):
        mode = ops.current_mode()
        adt = ops.act_storage(mode)
        for cin, cout in ((512, 512), (80, 512), (512, 80)):
            dy = torch.randn(R, cout, device="cuda").to(adt)
            xa = torch.relu(torch.randn(R, cin, device="cuda")).to(adt)
            dw = torch.zeros(5, cout, cin, device="cuda")
            auto = ops._split_k(5 * ops._tiles(cout, cin), R)
            row = []
            for sk in sorted({2, 3, 4, 5, 6, 8, 10, 12, 16, 19, 24, 32, auto}):
                us = timeit(lambda: check(L.dvae_conv5_wgrad(ptr(dy), ptr(xa), ptr(dw), R, nseg, cin, cout, sk,
                                                             ops._mflags(mode, dy, xa), stream()), "wgrad"))
                row.append(f"{sk}{'*' if sk == auto else ''}:{us:.0f}")
            print(f"{dtype} R={R} {cin}->{cout}  us by splits (* = ops._split_k):", " ".join(row), flush=True)
