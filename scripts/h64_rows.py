"""GPU box (dev library): the H = 64 bidirectional recurrence (N = 128 segments, T = 128) with 16 / 8 / 4 segments per
workgroup (DVAE_LSTM_H64_ROWS): time per launch of the forward and the backward pass, and that the results do not move."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) < 2:
    for rows in (16, 8, 4):
        env = dict(os.environ, DVAE_LSTM_H64_ROWS=str(rows),
                   DVAE_LIB_PATH=os.path.join(ROOT, "disentangle-vae-for-vc_amd", "libdvae_dev.so"))
        subprocess.run([sys.executable, os.path.abspath(__file__), str(rows)], env=env, check=True)
    sys.exit(0)
sys.path.insert(0, ROOT)
import hashlib
import torch
import dvae_amd  # noqa
from dvae_amd import ops
H, In, T, N = 64, 512, 128, 128
g = torch.Generator(device="cuda").manual_seed(3)
P = lambda *s: torch.nn.Parameter(torch.randn(*s, device="cuda", generator=g) * 0.1)
ps = [P(4 * H, In), P(4 * H, H), P(4 * H), P(4 * H), P(4 * H, In), P(4 * H, H), P(4 * H), P(4 * H)]
for p in ps:
    p.grad = torch.zeros_like(p)
x = torch.randn(T * N, In, device="cuda", generator=g).requires_grad_()
gh = torch.randn(T * N, 2 * H, device="cuda", generator=g)
def run():
    h = ops.LstmLayerFn.apply(x, T, N, *ps)
    h.backward(gh)
    return h
ops.set_deterministic(True)
h = run()
dig = hashlib.sha256(h.detach().cpu().numpy().tobytes() + x.grad.cpu().numpy().tobytes()).hexdigest()[:16]
ops.set_deterministic(False)
for _ in range(5):
    run()
torch.cuda.synchronize()
ops.prof_enable(2)
for _ in range(20):
    run()
torch.cuda.synchronize()
ms, n, fl = ops.prof_collect(); ops.prof_enable(0)
print(f"rows {sys.argv[1]:>2}: recurrence launches fwd + bwd {ms / 20 * 1e3:.1f} us per layer ({n // 20} launches)   digest(h, dx) {dig}")
