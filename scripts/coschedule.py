"""GPU-box experiment (eager, two HIP streams): the backward recurrence of an H=1024 LSTM layer (L2-bandwidth bound)
next to weight-gradient GEMMs (MFMA bound).  With DVAE_LSTM_MT5=1 the frame workgroup needs 78 KB of LDS and with
DVAE_GEMM_DYNLDS=14000 a GEMM workgroup 81 KB: one of each fits a CU.  usage: coschedule.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dvae_amd  # noqa
from dvae_amd import ops
T, N, H, In = 128, 128, 1024, 512
P = lambda *s: torch.nn.Parameter(torch.randn(*s, device="cuda") * 0.05)
ps = [P(4 * H, In), P(4 * H, H), P(4 * H), P(4 * H)]
for p in ps:
    p.grad = torch.zeros_like(p)
x = torch.randn(T * N, In, device="cuda")
gh = torch.randn(T * N, H, device="cuda")
dg, hh, gw = torch.randn(T * N, 4096, device="cuda"), torch.randn(T * N, 1024, device="cuda"), torch.zeros(4096, 1024, device="cuda")
side = torch.cuda.Stream()
def lstm():
    xx = x.clone().requires_grad_()
    h = ops.LstmLayerFn.apply(xx, T, N, *ps, None, None, None, None)
    h.backward(gh)
def gemms(n=4):
    for _ in range(n):
        ops.linear_wgrad_acc(dg, hh, gw)
def both():
    main = torch.cuda.current_stream()
    side.wait_stream(main)
    with torch.cuda.stream(side):
        gemms()
    lstm()
    main.wait_stream(side)
def timeit(fn, name):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10): fn()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 10 * 1e3
    print(f"{name:28s} {ms:7.3f} ms", flush=True)
    return ms
a = timeit(lstm, "lstm layer fwd+bwd alone")
b = timeit(gemms, "4 wgrad GEMMs alone")
c = timeit(both, "both, two streams")
print(f"sum {a + b:.3f}  concurrent {c:.3f}  saved {a + b - c:.3f} ms ({100 * (a + b - c) / (a + b):.0f} %)")
