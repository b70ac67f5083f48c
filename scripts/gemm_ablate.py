"""GPU box: the split-mode contraction kernel with parts of its k loop removed (libraries built with
-DDVAE_GEMM_ABL=<bits> under build_exp/, see gemm.hip) — where the time of a k-tile goes.  Results of ablated builds are wrong."""
import ctypes as C
import glob
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
dev = "cuda"
N, T = 128, 128
R = N * T
vp, i32, i64 = C.c_void_p, C.c_int, C.c_int64
X3 = 2
st = torch.cuda.current_stream().cuda_stream


def timeit(fn):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


x = torch.relu(torch.randn(R, 512, device=dev))
w = torch.randn(5, 512, 512, device=dev) * 0.05
b = torch.zeros(512, device=dev)
y = torch.empty(R, 512, device=dev)
dy = torch.randn(R, 512, device=dev)
dw = torch.zeros(5, 512, 512, device=dev)
a2 = torch.randn(R, 1024, device=dev)
w2 = torch.randn(4096, 1024, device=dev) * 0.05
c2 = torch.empty(R, 4096, device=dev)
libs = [os.path.join(ROOT, "disentangle-vae-for-vc_amd", "libdvae_hip.so")] + sorted(glob.glob(os.path.join(ROOT, "build_exp", os.environ.get("ABL_GLOB", "libabl_*.so"))))
ref = {}
for path in libs:
    L = C.CDLL(path)
    L.dvae_conv5_fwd.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, i32, vp]
    L.dvae_conv5_wgrad.argtypes = [vp, vp, vp, i32, i32, i32, i32, i32, i32, vp]
    L.dvae_gemm_f32.argtypes = [vp, vp, vp, vp, i32, i32, i32, i64, i64, i64, i32, i32, i32, i32, i32, i32, vp]
    p = lambda t: t.data_ptr()
    t1 = timeit(lambda: L.dvae_conv5_fwd(p(x), p(w), p(b), p(y), R, N, 512, 512, X3, st))
    t2 = timeit(lambda: L.dvae_conv5_wgrad(p(dy), p(x), p(dw), R, N, 512, 512, 3, X3, st))
    dw.zero_()
    L.dvae_conv5_wgrad(p(dy), p(x), p(dw), R, N, 512, 512, 3, X3, st)
    t3 = timeit(lambda: L.dvae_gemm_f32(p(a2), p(w2), p(c2), None, R, 4096, 1024, 1024, 1024, 4096, 1, 1, 0, 0, 1, X3, st))
    outs = {"y": y.clone(), "dw": dw.clone(), "c2": c2.clone()}
    if not ref:
        ref = outs
    dif = " ".join(f"{k}:{float((outs[k] - ref[k]).abs().max()):.1e}" for k in outs)
    f1, f3 = 2.0 * R * 512 * 512 * 5, 2.0 * R * 4096 * 1024
    print(f"{os.path.basename(path):18s} conv fwd {t1:7.1f} us ({f1 / t1 / 1e6:6.1f} TF/s)  conv wgrad {t2:7.1f} us ({f1 / t2 / 1e6:6.1f})  "
          f"nt 16384x4096x1024 {t3:7.1f} us ({f3 / t3 / 1e6:6.1f})  max|diff| {dif}", flush=True)
