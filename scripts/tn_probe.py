"""GPU box (under rocprofv3 --pmc): one weight-gradient shape of the bf16 mode, row-contiguous operands, k-splits into slabs,
20 launches.  DVAE_GEMM_256=0: the tall kernel, =2: the 256 x 256 LDS-DMA kernel (dev build)."""
import os
os.environ.setdefault("DVAE_LIB_PATH", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                     "disentangle-vae-for-vc_amd", "libdvae_dev.so"))
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dvae_amd  # noqa
from dvae_amd import ops
ops.set_compute_dtype("bf16")
M, N, K = 4096, 1024, 65536
g = torch.Generator(device="cuda").manual_seed(1)
a = (torch.rand(K, M, device="cuda", generator=g) * 2 - 1).bfloat16()
b = (torch.rand(K, N, device="cuda", generator=g) * 2 - 1).bfloat16()
c = torch.zeros(M, N, device="cuda")
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 20):
    ops.wgrad_gemm(a, b, c, None, M, N, K, M, N, False, False, 2, ops.MODE_BF16)
torch.cuda.synchronize()
print("done", os.environ.get("DVAE_GEMM_256"))
