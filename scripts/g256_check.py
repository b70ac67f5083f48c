"""GPU box: the 256 x 256 LDS-DMA kernel of the bf16 mode (csrc/gemm256.hip) — correctness against the 128 x 128 kernel
(bit for bit where the product is unsplit) and fp64, then timing of the shapes BASELINE configs[2] runs through it.
Needs the DEVELOPMENT build (DVAE_GEMM_256=0 switches the new kernel off: run once with, once without for the A/B).
usage: g256_check.py [check|time|both] [reps]"""
import os as _os
_os.environ.setdefault("DVAE_LIB_PATH", _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
                                                       "disentangle-vae-for-vc_amd", "libdvae_dev.so"))
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dvae_amd  # noqa
from dvae_amd import ops
from dvae_amd._lib import check, lib, ptr, stream

what = sys.argv[1] if len(sys.argv) > 1 else "both"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
L = lib()
ops.set_compute_dtype("bf16")
BF = ops.MODE_BF16
FL = ops.A_BF16 | ops.B_BF16
g = torch.Generator(device="cuda").manual_seed(5)
rnd = lambda *s: (torch.rand(*s, device="cuda", generator=g) * 2 - 1)
r16 = lambda x: x.bfloat16().float()
bad = 0


def report(name, ok, extra=""):
    global bad
    bad += 0 if ok else 1
    print(f"{'ok  ' if ok else 'FAIL'} {name} {extra}", flush=True)


class tall_only:
    """DVAE_GEMM_256=0 for the launches inside: the dev build reads the knob per call"""
    def __enter__(self):
        os.environ["DVAE_GEMM_256"] = "0"
    def __exit__(self, *a):
        os.environ["DVAE_GEMM_256"] = "2"      # 2: the 256 x 256 kernel for BOTH operand layouts (dev build)


def both(fn):
    """fn() with the tall kernel, then with the 256 x 256 kernel"""
    with tall_only():
        a = fn()
    return a, fn()


def gemm_pair(M, N, K, kc, epi=ops.EPI_STORE, sk=1):
    """the same product from fp32-stored (128 x 128 kernel) and bf16-stored operands (tall / 256 kernel)"""
    a, b = r16(rnd(M, K)), r16(rnd(K, N))
    A = a if kc else a.t().contiguous()
    B = b.t().contiguous() if kc else b
    lda, ldb = (K if kc else M), (K if kc else N)
    out = []
    for A_, B_, fl in ((A, B, 0), (A.bfloat16(), B.bfloat16(), FL)):
        C_ = torch.zeros(M, N, device="cuda")
        ops.gemm(A_, B_, C_, None, M, N, K, lda, ldb, N, kc, kc, 0, epi, sk, BF | fl)
        out.append(C_)
    return out[0], out[1], a, b


os.environ["DVAE_GEMM_256"] = "2"
if what in ("check", "both"):
    for (M, N, K, kc) in [(65536, 512, 512, True), (3880, 3848, 576, True), (4096, 4096, 1024, False),
                          (3880, 3848, 576, False), (65536, 256, 64 * 9, True), (16384, 4096, 512, True)]:
        c32, c16, a, b = gemm_pair(M, N, K, kc)
        ref = a[:512].double() @ b.double()
        err = float((c16[:512].double() - ref).abs().max())
        report(f"gemm {'nt' if kc else 'tn'} M={M} N={N} K={K}", torch.equal(c32, c16) and err < 1e-3,
               f"bitwise={torch.equal(c32, c16)} maxdiff={float((c32 - c16).abs().max()):.3e} err64={err:.2e}")
    # atomically accumulated k-splits (LSTM weight gradients): against fp64
    for (M, N, K, sk) in [(4096, 1024, 65536, 2), (2048, 512, 65280 // 64 * 64, 8)]:
        c32, c16, a, b = gemm_pair(M, N, K, False, ops.EPI_ATOMIC, sk)
        ref = a[:256].double() @ b.double()
        e16 = float((c16[:256].double() - ref).abs().max())
        e32 = float((c32[:256].double() - ref).abs().max())
        report(f"gemm tn atomic M={M} N={N} K={K} sk={sk}", e16 < 4 * max(e32, 1e-4), f"err64 {e16:.2e} (128 x 128 kernel {e32:.2e})")
    # the weight-gradient form with its k-splits in slabs: tall kernel against the 256 x 256 kernel (different split counts:
    # not bitwise), fp64, and run to run
    for (M, N, K, sk) in [(4096, 1024, 65536, 2), (2048, 512, 65280 // 64 * 64, 8)]:
        a, b = rnd(K, M).bfloat16(), rnd(K, N).bfloat16()

        def wg():
            g_ = torch.zeros(M, N, device="cuda")
            ops.wgrad_gemm(a, b, g_, None, M, N, K, M, N, False, False, sk, BF)
            return g_
        t_, n_ = both(wg)
        n2 = wg()
        ref = a[:, :256].double().t() @ b.double()
        e1 = float((n_[:256].double() - ref).norm() / ref.norm())
        report(f"wgrad tn slabs M={M} N={N} K={K}", torch.equal(n_, n2) and e1 < 5e-6 and float((t_ - n_).abs().max()) < 1e-2,
               f"relL2 vs fp64 {e1:.2e}, run-to-run bitwise {torch.equal(n_, n2)}, max |tall - 256| {float((t_ - n_).abs().max()):.2e}")
    # bias + activation epilogues
    M, N, K = 65536, 512, 512
    a, b, bias = r16(rnd(M, K)), r16(rnd(N, K)), rnd(N)
    for act in (0, 1, 2):
        outs = []
        for A_, B_, fl in ((a, b, 0), (a.bfloat16(), b.bfloat16(), FL)):
            C_ = torch.empty(M, N, device="cuda")
            ops.gemm(A_, B_, C_, bias, M, N, K, K, K, N, True, True, act, ops.EPI_STORE, 1, BF | fl)
            outs.append(C_)
        report(f"gemm nt bias act={act}", torch.equal(*outs))
    # convs: R = 65536 rows of N segments; bf16 operands, the tall kernel (DVAE_GEMM_256=0) against the 256 x 256 kernel
    for (R, N, Cin, Cout) in [(65536, 128, 512, 512), (65536 - 256, 256, 512, 512), (16384 * 3, 64, 512, 1024)]:
        x, wp, bias = rnd(R, Cin).bfloat16(), (rnd(5, Cout, Cin) * 0.1).bfloat16(), rnd(Cout)

        def fwd():
            y = torch.empty(R, Cout, device="cuda")
            check(L.dvae_conv5_fwd(ptr(x), ptr(wp), ptr(bias), ptr(y), R, N, Cin, Cout, BF | FL, stream()), "fwd")
            return y
        ys = both(fwd)
        # fp64 on the first 2 N + 64 rows (covers the zero padding above row 0)
        rows = 2 * N + 64
        xr = torch.zeros(rows + 4 * N, Cin, device="cuda", dtype=torch.float64)
        xr[2 * N:] = x[:rows + 2 * N].double()
        ref = sum(xr[tap * N: tap * N + rows] @ wp[tap].double().t() for tap in range(5)) + bias.double()
        err = float((ys[1][:rows].double() - ref).abs().max())
        report(f"conv fwd R={R} N={N} {Cin}->{Cout}", torch.equal(*ys) and err < 1e-3, f"bitwise vs tall={torch.equal(*ys)} err64={err:.2e}")
        for G in (1, 2):
            ws_b = L.dvae_bn_ws_bytes(R, Cout, G)

            def fwd_stats():
                y = torch.empty(R, Cout, device="cuda")
                ws = torch.zeros(ws_b // 8, device="cuda", dtype=torch.float64)
                check(L.dvae_conv5_fwd_stats(ptr(x), ptr(wp), ptr(bias), ptr(y), R, N, Cin, Cout, BF | FL, G, ptr(ws), stream()), "fwd_stats")
                return y, ws
            st = both(fwd_stats)
            report(f"conv fwd + BN stats G={G} R={R}", torch.equal(st[0][0], st[1][0]) and torch.equal(st[0][1], st[1][1]) and torch.equal(st[1][0], ys[1]))
        gy, wpt = rnd(R, Cout).bfloat16(), (rnd(5, Cin, Cout) * 0.1).bfloat16()

        def dgrad():
            dx = torch.empty(R, Cin, device="cuda")
            check(L.dvae_conv5_dgrad_t(ptr(gy), ptr(wpt), ptr(dx), R, N, Cin, Cout, BF | FL, stream()), "dgrad")
            return dx
        ds = both(dgrad)
        report(f"conv dgrad R={R} N={N}", torch.equal(*ds))

        def wgrad():
            dw = torch.zeros(5, Cout, Cin, device="cuda")
            check(L.dvae_conv5_wgrad(ptr(gy), ptr(x), ptr(dw), R, N, Cin, Cout, 6, BF | FL, stream()), "wgrad")
            return dw
        dws = both(wgrad)
        ref = torch.zeros(5, 64, Cin, device="cuda", dtype=torch.float64)
        for tap in range(5):
            sh = (tap - 2) * N
            xs = torch.zeros(R, Cin, device="cuda", dtype=torch.float64)
            if sh >= 0: xs[:R - sh] = x[sh:].double()
            else: xs[-sh:] = x[:R + sh].double()
            ref[tap] = gy[:, :64].double().t() @ xs
        e16 = float((dws[1][:, :64].double() - ref).abs().max())
        e32 = float((dws[0][:, :64].double() - ref).abs().max())
        report(f"conv wgrad R={R} N={N}", e16 < 4 * max(e32, 1e-3), f"err64 {e16:.2e} (tall kernel {e32:.2e})")
    # repeated launches: bit-identical run to run (a staging race would show here)
    M, N, K = 65536, 512, 4096
    a, b = rnd(M, K).bfloat16(), rnd(N, K).bfloat16()
    first = None
    same = True
    for i in range(30):
        C_ = torch.empty(M, N, device="cuda")
        ops.gemm(a, b, C_, None, M, N, K, K, K, N, True, True, 0, ops.EPI_STORE, 1, BF | FL)
        if first is None: first = C_.clone()
        else: same &= torch.equal(first, C_)
    report("30 launches of M=65536 N=512 K=4096 bitwise equal", same)
    print("CHECK", "PASSED" if bad == 0 else f"FAILED ({bad})", flush=True)

if what in ("time", "both"):
    def timeit1(name, fn, fl):
        for _ in range(10): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        return ms

    def timeit(name, fn, fl):
        res = []
        for rnd_ in range(2):          # interleaved rounds, one process (rule 24)
            with tall_only():
                a = timeit1(name, fn, fl)
            res.append((a, timeit1(name, fn, fl)))
        a, b = min(r[0] for r in res), min(r[1] for r in res)
        print(f"{name:44s} tall {a * 1e3:8.1f} us {fl / a / 1e9:7.1f} TF/s | 256 {b * 1e3:8.1f} us {fl / b / 1e9:7.1f} TF/s  ({a / b:.2f}x)", flush=True)

    R, N = 65536, 256
    t16 = lambda *s: rnd(*s).bfloat16()
    x, wp, b, y = t16(R, 512), t16(5, 512, 512), rnd(512), torch.empty(R, 512, device="cuda")
    timeit("conv fwd 512->512 R=65536", lambda: check(L.dvae_conv5_fwd(ptr(x), ptr(wp), ptr(b), ptr(y), R, N, 512, 512, BF | FL, stream()), ""), 10.0 * R * 512 * 512)
    ws = torch.zeros(L.dvae_bn_ws_bytes(R, 512, 2) // 8, device="cuda", dtype=torch.float64)
    timeit("conv fwd + BN stats", lambda: check(L.dvae_conv5_fwd_stats(ptr(x), ptr(wp), ptr(b), ptr(y), R, N, 512, 512, BF | FL, 2, ptr(ws), stream()), ""), 10.0 * R * 512 * 512)
    timeit("conv dgrad", lambda: check(L.dvae_conv5_dgrad_t(ptr(x), ptr(wp), ptr(y), R, N, 512, 512, BF | FL, stream()), ""), 10.0 * R * 512 * 512)
    dw = torch.zeros(5, 512, 512, device="cuda")
    timeit("conv wgrad sk=6", lambda: check(L.dvae_conv5_wgrad(ptr(x), ptr(x), ptr(dw), R, N, 512, 512, 6, BF | FL, stream()), ""), 10.0 * R * 512 * 512)
    slab = torch.empty(16 * dw.numel(), device="cuda")

    def conv_wgrad_slabs():
        n = L.dvae_conv5_wgrad_slabs(ptr(x), ptr(x), ptr(dw), ptr(slab), dw.numel(), 16, R, N, 512, 512, ops.EPI_ACCUM, 6, BF | FL, stream())
        check(L.dvae_slab_sum(ptr(dw), ptr(slab), dw.numel(), n, dw.numel(), 0, 1, stream()), "")
    timeit("conv wgrad, k-splits into slabs + sum", conv_wgrad_slabs, 10.0 * R * 512 * 512)
    for (M, Nn, K, kc, epi, sk) in [(65536, 4096, 1024, True, ops.EPI_STORE, 1), (65536, 1024, 4096, True, ops.EPI_STORE, 1),
                                    (65536, 4096, 512, True, ops.EPI_STORE, 1), (65536, 512, 4096, True, ops.EPI_STORE, 1),
                                    (4096, 1024, 65536, False, ops.EPI_ATOMIC, 2), (4096, 512, 65536, False, ops.EPI_ATOMIC, 4),
                                    (2048, 512, 65280, False, ops.EPI_ATOMIC, 8), (65536, 512, 512, True, ops.EPI_STORE, 1),
                                    (8192, 8192, 8192, True, ops.EPI_STORE, 1)]:
        A = t16(M, K) if kc else t16(K, M)
        Bm = t16(Nn, K) if kc else t16(K, Nn)
        Cm = torch.zeros(M, Nn, device="cuda")
        lda, ldb = (K if kc else M), (K if kc else Nn)
        if kc:
            timeit(f"gemm nt M={M} N={Nn} K={K} sk={sk}",
                   lambda: ops.gemm(A, Bm, Cm, None, M, Nn, K, lda, ldb, Nn, kc, kc, 0, epi, sk, BF | FL), 2.0 * M * Nn * K)
        else:      # weight-gradient shapes: k-splits into slabs, summed right behind (no optimiser owns Cm)
            timeit(f"wgrad tn M={M} N={Nn} K={K} sk={sk} (slabs + sum)",
                   lambda: ops.wgrad_gemm(A, Bm, Cm, None, M, Nn, K, lda, ldb, False, False, sk, BF), 2.0 * M * Nn * K)
