"""GPU-box microbench: every contraction shape of the B=64,T=128 step through the C ABI, TF/s per shape."""
import os as _os
# needs the DEVELOPMENT build of the library (csrc/build.sh dev): probes / environment knobs / timelines are not in the product
_os.environ.setdefault("DVAE_LIB_PATH", _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
                                                       "disentangle-vae-for-vc_amd", "libdvae_dev.so"))
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dvae_amd  # noqa: F401
from dvae_amd import ops
from dvae_amd._lib import check, lib, ptr, stream

R, N = 16384, 128
dev = "cuda"


def t(*s):
    return torch.randn(*s, device=dev)


def timeit(fn, flops, name, reps=100):
    # sustained: a burst of a few launches runs ~13 % slower (clocks still ramping) and says nothing about the step
    for _ in range(30):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"{name:50s} {ms * 1e3:9.1f} us  {flops / ms / 1e9:7.1f} TF/s", flush=True)
    return ms


L = lib()
MODE = ops.current_mode()        # DVAE_COMPUTE_DTYPE=fp32|fp32x3|bf16
print("compute mode:", ops.get_compute_dtype(), flush=True)
tot = 0.0
for cin, cout, cnt in ((80, 512, 2), (512, 512, 8), (512, 80, 1)):
    x, wp, b, y = t(R, cin), t(5, cout, cin), t(cout), torch.empty(R, cout, device=dev)
    fl = 2.0 * R * cin * cout * 5
    tot += cnt * timeit(lambda: check(L.dvae_conv5_fwd(ptr(x), ptr(wp), ptr(b), ptr(y), R, N, cin, cout, MODE, stream()), ""),
                        fl, f"conv_fwd {cin}->{cout} x{cnt}")
    dx = torch.empty(R, cin, device=dev)
    wpt = t(5, cin, cout)
    tot += cnt * timeit(lambda: check(L.dvae_conv5_dgrad_t(ptr(y), ptr(wpt), ptr(dx), R, N, cin, cout, MODE, stream()), ""),
                        fl, f"conv_dgrad {cin}->{cout} x{cnt}")
    dwp = torch.zeros(5, cout, cin, device=dev)
    sk = ops._split_k(5 * ops._tiles(cout, cin), R)
    tot += cnt * timeit(lambda: check(L.dvae_conv5_wgrad(ptr(y), ptr(x), ptr(dwp), R, N, cin, cout, sk, MODE, stream()), ""),
                        fl, f"conv_wgrad {cin}->{cout} sk{sk} x{cnt}")
for M, K, No, cnt, nm in ((R, 512, 256, 2, "enc_lstm0 inproj"), (R, 128, 256, 2, "enc_lstm1 inproj"),
                          (R, 128, 2048, 1, "dec_lstm1 inproj"), (R, 512, 4096, 1, "dec_lstm2.0 inproj"),
                          (R, 1024, 4096, 1, "dec_lstm2.1 inproj"), (R, 1024, 80, 1, "dec_linear2"),
                          (N, 16384, 2048, 1, "enc_linear"), (N, 2048, 16384, 1, "dec_pre_linear2"),
                          (N, 32, 2048, 1, "dec_pre_linear1"), (N, 2048, 56, 1, "content")):
    x, w, b = t(M, K), t(No, K), t(No)
    fl = 2.0 * M * K * No
    tot += cnt * timeit(lambda: ops.linear_fwd(x, w, b), fl, f"fwd   {nm} [{M}x{K}]x[{No}] x{cnt}")
    dy = t(M, No)
    tot += cnt * timeit(lambda: ops.linear_dgrad(dy, w), fl, f"dgrad {nm} x{cnt}")
    gw = torch.zeros(No, K, device=dev)
    tot += cnt * timeit(lambda: ops.linear_wgrad_acc(dy, x, gw), fl, f"wgrad {nm} x{cnt}")
for H, cnt in ((64, 4), (512, 1), (1024, 2)):
    dg, h, gw = t(R, 4 * H), t(R, H), torch.zeros(4 * H, H, device=dev)
    fl = 2.0 * (R - N) * 4 * H * H
    tot += cnt * timeit(lambda: ops.linear_wgrad_acc(dg, h, gw, rows=R - N), fl, f"wgrad W_hh H={H} x{cnt}")
print("sum of contraction time per step (ms):", tot)
