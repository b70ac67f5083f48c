"""GPU box (dev library): the LDS-DMA kernel on v_mfma_f32_16x16x32_bf16 (DVAE_GEMM_256_SHAPE=16, experiment) against the shipped
32x32x16 form: fp64 check, run-to-run, interleaved timing."""
import os as _os
_os.environ.setdefault("DVAE_LIB_PATH", _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
                                                       "disentangle-vae-for-vc_amd", "libdvae_dev.so"))
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dvae_amd  # noqa
from dvae_amd import ops
from dvae_amd._lib import check, lib, ptr, stream
L = lib()
ops.set_compute_dtype("bf16")
BF, FL = ops.MODE_BF16, ops.A_BF16 | ops.B_BF16
os.environ["DVAE_GEMM_256"] = "2"
g = torch.Generator(device="cuda").manual_seed(5)
rnd = lambda *s: (torch.rand(*s, device="cuda", generator=g) * 2 - 1)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 60


def shape(v):
    os.environ["DVAE_GEMM_256_SHAPE"] = str(v)


for (M, N, K) in [(65536, 512, 512), (3880, 3848, 576), (8192, 8192, 1024)]:
    a, b, bias = rnd(M, K).bfloat16(), rnd(N, K).bfloat16(), rnd(N)
    outs = {}
    for v in (32, 16, 16):
        shape(v)
        c = torch.empty(M, N, device="cuda")
        ops.gemm(a, b, c, bias, M, N, K, K, K, N, True, True, 1, ops.EPI_STORE, 1, BF | FL)
        outs.setdefault(v, []).append(c)
    ref = torch.relu(a[:256].double() @ b.double().t() + bias.double())
    e16 = float((outs[16][0][:256].double() - ref).norm() / ref.norm())
    e32 = float((outs[32][0][:256].double() - ref).norm() / ref.norm())
    print(f"M={M} N={N} K={K}: relL2 vs fp64: 16x16x32 {e16:.2e}, 32x32x16 {e32:.2e}; 16x16 run-to-run bitwise "
          f"{torch.equal(outs[16][0], outs[16][1])}; max |16 - 32| {float((outs[16][0] - outs[32][0]).abs().max()):.2e}", flush=True)
R, Ns = 65536, 256
x, wp, bb, y = rnd(R, 512).bfloat16(), (rnd(5, 512, 512) * 0.1).bfloat16(), rnd(512), torch.empty(R, 512, device="cuda")
ys = {}
for v in (32, 16):
    shape(v)
    check(L.dvae_conv5_fwd(ptr(x), ptr(wp), ptr(bb), ptr(y), R, Ns, 512, 512, BF | FL, stream()), "")
    ys[v] = y.clone()
print("conv fwd: max |16 - 32|", float((ys[16] - ys[32]).abs().max()), "scale", float(ys[32].abs().max()), flush=True)


def timeit(fn):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


cases = [("conv fwd 512->512 R=65536", lambda: check(L.dvae_conv5_fwd(ptr(x), ptr(wp), ptr(bb), ptr(y), R, Ns, 512, 512, BF | FL, stream()), ""), 10.0 * R * 512 * 512)]
for (M, N, K) in [(65536, 4096, 1024), (65536, 1024, 4096), (65536, 512, 4096), (8192, 8192, 8192)]:
    A, Bm, Cm = rnd(M, K).bfloat16(), rnd(N, K).bfloat16(), torch.empty(M, N, device="cuda")
    cases.append((f"gemm nt M={M} N={N} K={K}", (lambda A=A, Bm=Bm, Cm=Cm, M=M, N=N, K=K: ops.gemm(A, Bm, Cm, None, M, N, K, K, K, N, True, True, 0, ops.EPI_STORE, 1, BF | FL)), 2.0 * M * N * K))
for name, fn, fl in cases:
    res = {32: [], 16: []}
    for _ in range(2):
        for v in (32, 16):
            shape(v)
            res[v].append(timeit(fn))
    a, b = min(res[32]), min(res[16])
    print(f"{name:36s} 32x32x16 {a * 1e3:8.1f} us {fl / a / 1e9:7.1f} TF/s | 16x16x32 {b * 1e3:8.1f} us {fl / b / 1e9:7.1f} TF/s  ({a / b:.2f}x)", flush=True)
