"""Aggregate rocprofv3 --pmc counter_collection CSVs per kernel family and write profiles/pmc_traffic.json.
usage: pmc_summary.py <dir with pass sub-dirs> <n_steps> [tag] [traffic.json]   (prints a markdown summary)

Counters per MI355X_MICROARCH.md: FETCH_SIZE doubled (gfx950 reports 1/2 of a wide coalesced read), WRITE_SIZE as is; both
count L2 <-> fabric traffic (Infinity-Cache hits included).  MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x
1024 SIMDs); GRBM_GUI_ACTIVE over-counts on dispatches shorter than ~0.3 ms, so the LSTM frame kernels read low."""
import collections
import csv
import glob
import hashlib
import json
import os
import sys

root, steps = sys.argv[1], float(sys.argv[2])      # steps: fallback only — the real count is the number of Adam launches
tag = sys.argv[3] if len(sys.argv) > 3 else "pmc"
traffic_json = sys.argv[4] if len(sys.argv) > 4 else None
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
fam = lambda n: ("gemm_f32_kernel" if ("gemm_f32" in n or "gemm_x3" in n) else "lstm_step_* (H>=512, per frame)" if "lstm_step" in n else "lstm_pers_* (persistent)" if "lstm_pers" in n else
                 "lstm_seq_*_h64" if "lstm_seq" in n else "adam" if "adam" in n else "bn_*" if "bn_" in n else
                 "colsum" if "colsum" in n else "repack_all" if "repack" in n else "other")
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = fam(r["Kernel_Name"])
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[k][r["Counter_Name"]] += 1
# steps actually profiled = dispatches of the optimizer kernel (one per step) in one pass: the bench command runs warm-up,
# timed AND the step()-timing steps, so a count passed on the command line was wrong by 5/3 in round 3
n_adam = collections.Counter()
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "adam_dev_kernel" in r["Kernel_Name"]:
            n_adam[(f, r["Counter_Name"])] += 1
if n_adam:
    steps = float(max(n_adam.values()))
# algorithmic bytes per launch and instantiation, from a bench.py line of the same build (profile_round.sh: alg.json)
alg = {}
try:
    line = [l for l in open(os.path.join(root, "alg.json")) if l.lstrip().startswith("{")][-1]
    for it in json.loads(line)["roofline"]["instantiations"]:
        import re as _re
        name = it["kernel"].split("<")[0]
        kv = dict(_re.findall(r"(\w+)=(\d+)", it["kernel"]))
        b = lambda k: "true" if kv.get(k, "0") != "0" else "false"
        if name == "gemm_f32_kernel":
            key = f"{name}<{b('A_KC')}, {b('B_KC')}, {kv['NTW']}, {kv['BK']}, {kv['WG']}, {kv['MODE']}, {b('BNS')}, {b('A16')}, {b('B16M')}>"
        else:
            key = f"{name}<{b('A_KC')}, {b('B_KC')}, {b('BNS')}"
        e = alg.setdefault(key, [0.0, 0.0])
        e[0] += it["algorithmic_bytes_per_launch"] * it["launches_per_step"]
        e[1] += it["launches_per_step"]
except Exception as e:      # the table is printed without the column
    print(f"(no algorithmic bytes: {e!r})", file=sys.stderr)
print(f"# {tag} PMC summary (rocprofv3 --pmc, three separate passes over `bench.py --steps 2 --warmup 1 --graph 0`; counters "
      f"summed per kernel family over the {steps:.0f} steps run, divided by {steps:.0f})\n")
print(__doc__.split("\n\n", 1)[1] + "\n")
print("(gemm_f32_kernel = the contraction family: gemm_f32_kernel<...> and gemm_x3_tall_kernel<...>)\n")
print("| kernel family | launches/step | MFMA busy | fabric-side read MB/step (2 x FETCH_SIZE KB) | write MB/step (WRITE_SIZE KB) |")
print("|---|---|---|---|---|")
for k in sorted(acc, key=lambda k: -acc[k].get("GRBM_GUI_ACTIVE", 0)):
    a = acc[k]
    n = max(cnt[k].values()) / steps
    busy = a.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / max(1.0, a.get("GRBM_GUI_ACTIVE", 0) / 8 * 1024)
    print(f"| {k} | {n:.0f} | {busy:.2f} | {2 * a.get('FETCH_SIZE', 0) / 1024 / steps:.0f} | {a.get('WRITE_SIZE', 0) / 1024 / steps:.0f} |")
# per instantiation of the contraction family (template arguments as rocprofv3 prints them)
import re
inst = collections.defaultdict(lambda: collections.defaultdict(float))
icnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "gemm_f32" in n or "gemm_x3" in n:
            k = re.sub(r"\(anonymous namespace\)::|^void |\(.*$", "", n)[:70]
            inst[k][r["Counter_Name"]] += float(r["Counter_Value"])
            icnt[k][r["Counter_Name"]] += 1
if inst:
    print("\nContraction family by instantiation (fabric-side MB per LAUNCH: 2 x FETCH_SIZE + WRITE_SIZE):\n")
    print("| instantiation | launches/step | read MB/launch | write MB/launch | algorithmic MB/launch (operands + result, each once) | fabric / algorithmic |")
    print("|---|---|---|---|---|---|")
    for k in sorted(inst, key=lambda k: -(2 * inst[k].get("FETCH_SIZE", 0) + inst[k].get("WRITE_SIZE", 0))):
        n = max(1, icnt[k].get("FETCH_SIZE", 0))
        rd = 2 * inst[k].get('FETCH_SIZE', 0) / 1024 / n
        wr = inst[k].get('WRITE_SIZE', 0) / 1024 / max(1, icnt[k].get('WRITE_SIZE', 0))
        a = next((v for kk, v in alg.items() if k.startswith(kk) or kk.startswith(k)), None)
        am = a[0] / max(1e-9, a[1]) / 1e6 if a else None
        print(f"| {k} | {n / steps:.0f} | {rd:.1f} | {wr:.1f} | " + (f"{am:.1f} | {(rd + wr) / am:.2f} |" if am else "- | - |"))
g = acc.get("gemm_f32_kernel")
if g and g.get("FETCH_SIZE") and g.get("WRITE_SIZE"):
    launches = cnt["gemm_f32_kernel"]["FETCH_SIZE"] / steps
    rd, wr = 2 * g["FETCH_SIZE"] * 1024 / steps, g["WRITE_SIZE"] * 1024 / steps
    h = hashlib.sha256()
    for f in ("gemm.hip", "gemm256.hip", "gemm_common.h", "common.h"):      # = bench.kernel_source_hash()
        h.update(open(os.path.join(ROOT, "disentangle-vae-for-vc_amd", "csrc", f), "rb").read())
    dtype = os.environ.get("DVAE_COMPUTE_DTYPE", "fp32x3")
    out = {"source": f"profiles/{tag}_pmc_summary.md (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; FETCH_SIZE x2 per "
                     "the gfx950 correction), scripts/profile_round.sh",
           "kernel_source_sha16": h.hexdigest()[:16], "workload": f"B=64,T=128,{dtype}",
           "gemm_f32_kernel": {"launches_per_step": launches, "read_bytes_per_step": rd, "write_bytes_per_step": wr,
                               "traffic_bytes_per_launch": (rd + wr) / launches}}
    json.dump(out, open(traffic_json or os.path.join(ROOT, "profiles", "pmc_traffic.json"), "w"), indent=1)
    print(f"\ngemm_f32_kernel: ({rd / 1e6:.0f} + {wr / 1e6:.0f}) MB / {launches:.0f} launches = {(rd + wr) / launches / 1e6:.1f} MB of "
          "fabric traffic per launch (written to profiles/pmc_traffic.json)")
