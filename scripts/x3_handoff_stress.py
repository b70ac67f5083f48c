"""GPU box: the fp32x3 persistent recurrences (forward + backward) against the per-frame kernels under foreign HBM traffic,
many rounds; prints which tensor / frame / row group went wrong when one does.
usage: x3_handoff_stress.py H T N rounds      (DVAE_LIB_PATH + DVAE_PERS_X3_MT1 / DVAE_PERS_BWD_KSPLIT pick the kernels)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import dvae_amd  # noqa: F401
from dvae_amd import _lib, ops
from dvae_amd.derived import lstm_local
from test_hip_lstm_pers import _x3_pass

H, T, N, rounds = (int(v) for v in sys.argv[1:5])
env = (_lib, ops, lstm_local)
ref = _x3_pass(env, H, T, N, pers=False)
side = torch.cuda.Stream()
a = torch.empty(1 << 28, device="cuda", dtype=torch.float32)
b = torch.empty_like(a)
bad = 0
for rnd in range(rounds):
    with torch.cuda.stream(side):
        for _ in range(rnd % 5):
            b.copy_(a)
    got = _x3_pass(env, H, T, N, pers=True)
    for name, x, y in zip(("gates", "c", "h", "dgates"), got, ref):
        err = (x - y).abs()
        tol = 2e-5 * float(y.abs().max())
        if not torch.isfinite(x).all() or float(err.max()) > tol:
            bad += 1
            e = err.reshape(T, N, -1)
            frames = (e.amax(dim=(1, 2)) > tol).nonzero().flatten().tolist()
            rows = (e.amax(dim=(0, 2)) > tol).nonzero().flatten().tolist()
            cols = (e.amax(dim=(0, 1)) > tol).nonzero().flatten().tolist()
            f0 = frames[0] if frames else 0
            for nm2, x2, y2 in zip(("gates", "c", "h"), got[:3], ref[:3]):
                for ff in (f0 - 1, f0, f0 + 1):
                    if 0 <= ff < T:
                        e2 = (x2 - y2).abs().reshape(T, N, -1)[ff]
                        tol2 = 2e-5 * float(y2.abs().max())
                        bc = (e2.amax(dim=0) > tol2).nonzero().flatten().tolist()
                        br = (e2.amax(dim=1) > tol2).nonzero().flatten().tolist()
                        print(f"   {nm2} frame {ff}: bad cols {len(bc)} {bc[:4]}..{bc[-2:]}, bad rows {br}, max {float(e2.max()):.3e}")
            print(f"round {rnd} ({rnd % 5} GiB): {name} max {float(err.max()):.3e}; frames {frames[:6]}..({len(frames)}), rows "
                  f"{rows[:8]}..({len(rows)}), cols {cols[:8]}..({len(cols)})", flush=True)
            break
torch.cuda.synchronize()
print(f"H={H} T={T} N={N}: {bad} bad rounds of {rounds}")
