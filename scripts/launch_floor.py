"""GPU-box: per-launch floor of back-to-back dependent launches (eager and graph)."""
import os as _os
# needs the DEVELOPMENT build of the library (csrc/build.sh dev): probes / environment knobs / timelines are not in the product
_os.environ.setdefault("DVAE_LIB_PATH", _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
                                                       "disentangle-vae-for-vc_amd", "libdvae_dev.so"))
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dvae_amd  # noqa
from dvae_amd._lib import lib, check, stream
L = lib()
sink = torch.zeros(4, device="cuda")
for blocks, threads, lds in ((256, 512, 0), (256, 512, 80000), (512, 256, 40000), (64, 256, 0), (16, 512, 0)):
    n = 256
    def run(): check(L.dvae_probe_launches(n, blocks, threads, lds, sink.data_ptr(), stream()), "probe")
    run(); torch.cuda.synchronize()
    t0 = time.perf_counter(); run(); torch.cuda.synchronize(); eager = (time.perf_counter() - t0) / n * 1e6
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g): run()
    g.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): g.replay()
    torch.cuda.synchronize(); gr = (time.perf_counter() - t0) / 5 / n * 1e6
    print(f"blocks={blocks} threads={threads} lds={lds}: eager {eager:.2f} us/launch, graph {gr:.2f} us/launch")
