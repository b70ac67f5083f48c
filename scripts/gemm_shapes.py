"""GPU box: every contraction launch of one train step (B=64, T=128, fp32x3 unless DVAE_COMPUTE_DTYPE says otherwise) with
its shape, timed one by one (events around each launch of an eager step; 3 steps, last one reported).  Sorted by time."""
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from dvae_amd import _lib, ops
from dvae_amd.data import SyntheticPairs

B, T = int(os.environ.get("B", 64)), int(os.environ.get("T", 128))
dtype = os.environ.get("DVAE_COMPUTE_DTYPE", "fp32x3")
dev = torch.device("cuda", 0)
w = bench.build_trainer(dev, B, T, dtype)
x1, x2, spk = SyntheticPairs(B, T, n_speakers=10, seed=1234, device=dev).batch()
L = _lib.lib()
rec = []
names = ["dvae_gemm_f32", "dvae_conv5_fwd", "dvae_conv5_fwd_stats", "dvae_conv5_wgrad", "dvae_conv5_dgrad_t",
         # round 6: the k-split forms (their slab sums are separate launches: dvae_slab_sum / dvae_slab_fold, listed too)
         "dvae_gemm_f32_slabs", "dvae_gemm_f32_batched_slabs", "dvae_gemm_f32_batched", "dvae_conv5_wgrad_slabs",
         "dvae_conv5_fwd_slabs", "dvae_conv5_dgrad_t_slabs", "dvae_slab_sum", "dvae_slab_fold"]
orig = {n: getattr(L, n) for n in names}


def wrap(name):
    fn = orig[name]

    def f(*a):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = fn(*a)
        e1.record()
        if name == "dvae_gemm_f32_slabs":
            M, N, K, a_kc, b_kc, sk = a[7], a[8], a[9], a[13], a[14], a[16]
            key, fl = f"gemm_slabs M={M} N={N} K={K} akc={a_kc} bkc={b_kc} sk={sk}->{rc}", 2.0 * M * N * K
        elif name in ("dvae_gemm_f32_batched_slabs", "dvae_gemm_f32_batched"):
            o = 3 if name.endswith("slabs") else 0
            nb, M, N, K = a[3], a[4 + o], a[5 + o], a[6 + o]
            key, fl = f"{name[5:]} x{nb} M={M} N={N} K={K}", 2.0 * nb * M * N * K
        elif name == "dvae_conv5_wgrad_slabs":
            R, Cin, Cout, sk = a[6], a[8], a[9], a[11]
            key, fl = f"conv_wgrad_slabs R={R} Cin={Cin} Cout={Cout} sk={sk}->{rc}", 10.0 * R * Cin * Cout
        elif name in ("dvae_conv5_fwd_slabs", "dvae_conv5_dgrad_t_slabs"):
            o = 1 if name == "dvae_conv5_fwd_slabs" else 0
            R, Cin, Cout = a[6 + o], a[8 + o], a[9 + o]
            key, fl = f"{name[5:]} R={R} Cin={Cin} Cout={Cout} splits={rc}", 10.0 * R * Cin * Cout
        elif name == "dvae_slab_sum":
            key, fl = f"slab_sum n={a[4]} slabs={a[3]}", 0.0
        elif name == "dvae_slab_fold":
            key, fl = f"slab_fold entries={a[1]}", 0.0
        elif name == "dvae_gemm_f32":
            M, N, K, a_kc, b_kc, act, epi, sk = a[4], a[5], a[6], a[10], a[11], a[12], a[13], a[14]
            md = a[15]
            key, fl = f"gemm M={M} N={N} K={K} akc={a_kc} bkc={b_kc} epi={epi} sk={sk} mode={md & 0xff if md >= 0 else md} a16={(md >> 8) & 1 if md >= 0 else 0} b16={(md >> 9) & 1 if md >= 0 else 0}", 2.0 * M * N * K
        elif name == "dvae_conv5_wgrad":
            R, N_, Cin, Cout, sk = a[3], a[4], a[5], a[6], a[7]
            key, fl = f"conv_wgrad R={R} Cin={Cin} Cout={Cout} sk={sk}", 10.0 * R * Cin * Cout
        else:
            R, N_, Cin, Cout = a[4], a[5], a[6], a[7]
            if name == "dvae_conv5_dgrad_t":
                R, N_, Cin, Cout = a[3], a[4], a[5], a[6]
            key, fl = f"{name[5:]} R={R} Cin={Cin} Cout={Cout}", 10.0 * R * Cin * Cout
        rec.append((key, fl, e0, e1))
        return rc
    return f


for it in range(3):
    if it == 2:
        for n in names:
            setattr(L, n, wrap(n))
    w.step(x1, x2, spk, train=True)
torch.cuda.synchronize()
agg = defaultdict(lambda: [0, 0.0, 0.0])
for key, fl, e0, e1 in rec:
    a = agg[key]
    a[0] += 1
    a[1] += e0.elapsed_time(e1)
    a[2] += fl
tot = sum(a[1] for a in agg.values())
print(f"{len(rec)} launches, {tot:.2f} ms (event-timed one by one: includes launch gaps)")
for key, (n, ms, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{ms:7.3f} ms  x{n:2d}  {fl / ms / 1e9:7.1f} TF/s  {key}")
