"""GPU box, DEV library with DVAE_PERS_X3_MT1=3: the 16-row forward form of the fp32x3 persistent recurrence (H = 512) under
foreign HBM traffic; when a round comes back wrong, says WHAT the consumers must have read for the bad rows of the first
bad frame: h[t-1] of the right row (then the fault is elsewhere), of a stale frame (t-3: the same ring slot), zeros, or of
another row.  usage: x3_fwd16_diag.py [rounds]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("DVAE_LIB_PATH", os.path.join(ROOT, "disentangle-vae-for-vc_amd", "libdvae_dev.so"))
os.environ.setdefault("DVAE_PERS_X3_MT1", "3")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import dvae_amd  # noqa: F401
from dvae_amd import _lib, ops
from dvae_amd.derived import lstm_local
from test_hip_lstm_pers import _x3_pass

H, T, N = 512, 96, 128
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 60
env = (_lib, ops, lstm_local)
ref = [t.clone() for t in _x3_pass(env, H, T, N, pers=False)]
g = torch.Generator(device="cuda").manual_seed(1000)
w_hh = (torch.rand(4 * H, H, generator=g, device="cuda") * 2 - 1) / H ** 0.5
pre = (torch.rand(T * N, 4 * H, generator=g, device="cuda") * 2 - 1).reshape(T, N, 4 * H).double()
ref_h = ref[2].reshape(T, N, H).double()
W = w_hh.double()


def act(p):
    i, f, gg, o = p.split(H, dim=-1)
    return torch.cat([torch.sigmoid(i), torch.sigmoid(f), torch.tanh(gg), torch.sigmoid(o)], -1)


side = torch.cuda.Stream()
a = torch.empty(1 << 28, device="cuda", dtype=torch.float32)
b = torch.empty_like(a)
bad = 0
for rnd in range(rounds):
    with torch.cuda.stream(side):
        for _ in range(rnd % 5):
            b.copy_(a)
    got = _x3_pass(env, H, T, N, pers=True)
    gg_, gh_ = got[0].reshape(T, N, 4 * H).double(), got[2].reshape(T, N, H).double()
    e = (got[0] - ref[0]).abs().reshape(T, N, -1)
    tol = 2e-5 * float(ref[0].abs().max())
    if float(e.max()) <= tol:
        continue
    bad += 1
    frames = (e.amax(dim=(1, 2)) > tol).nonzero().flatten().tolist()
    f0 = frames[0]
    rows = (e[f0].amax(dim=1) > tol).nonzero().flatten().tolist()
    print(f"round {rnd} ({rnd % 5} GiB): first bad frame {f0} of {len(frames)}, bad rows {rows}", flush=True)
    n = rows[0]
    cands = {"h[t-1] same row (as stored)": gh_[f0 - 1, n], "ref h[t-1] same row": ref_h[f0 - 1, n], "zeros": torch.zeros(H, device="cuda", dtype=torch.float64)}
    for back in (2, 3, 4, 5):
        if f0 - back >= 0:
            cands[f"h[t-{back}] same row"] = ref_h[f0 - back, n]
    for dn in (-12, -8, -4, -1, 1, 4, 16, -16, 32, 64):
        if 0 <= n + dn < N:
            cands[f"h[t-1] row {n + dn:+d}".replace(f"{n + dn:+d}", str(n + dn))] = ref_h[f0 - 1, n + dn]
    res = []
    for name, hin in cands.items():
        sim = act(pre[f0, n] + hin @ W.t())
        res.append((float((sim - gg_[f0, n]).abs().max()), name))
    res.sort()
    for err, name in res[:4]:
        print(f"     consumers' input = {name:32s}: max |simulated - got gates| {err:.3e}")
    # per k-chunk: which 32-unit chunks of h were not the right ones?  solve for the input from the gates is not possible; compare
    # instead the stored h[t-1] (what the producers wrote to h_out) with the reference
    print(f"     stored h[{f0 - 1}] rows {rows}: max |stored - ref| {float((gh_[f0 - 1, rows] - ref_h[f0 - 1, rows]).abs().max()):.3e}")
    if bad >= 4:
        break
torch.cuda.synchronize()
print(f"{bad} bad rounds")
