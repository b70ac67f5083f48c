"""GPU-box experiment: two independent H=1024 LSTM recurrences, sequential vs on two streams."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dvae_amd  # noqa
from dvae_amd import ops
H, In, T, N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024, 1024, 128, 128
P = lambda *s: torch.randn(*s, device="cuda") * 0.05
def mk():
    return [P(4 * H, In), P(4 * H, H), P(4 * H), P(4 * H)] + [None] * 4, torch.randn(T * N, In, device="cuda")
(p1, x1), (p2, x2) = mk(), mk()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def seq():
    with torch.no_grad():
        ops.LstmLayerFn.apply(x1, T, N, *p1); ops.LstmLayerFn.apply(x2, T, N, *p2)
def par():
    with torch.no_grad():
        s1.wait_stream(torch.cuda.current_stream()); s2.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s1): ops.LstmLayerFn.apply(x1, T, N, *p1)
        with torch.cuda.stream(s2): ops.LstmLayerFn.apply(x2, T, N, *p2)
        torch.cuda.current_stream().wait_stream(s1); torch.cuda.current_stream().wait_stream(s2)
for name, fn in (("sequential", seq), ("two streams", par)):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g): fn()
    g.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): g.replay()
    torch.cuda.synchronize()
    print(f"H={H} {name}: {(time.perf_counter()-t0)/5*1e3:.3f} ms for 2 layers fwd (incl. 2 in-proj GEMMs)")
