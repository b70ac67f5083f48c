"""Kernel statistics from a rocprofv3 rocpd database (--kernel-trace --stats ... -o X  ->  X_results.db).

python scripts/rocpd_stats.py <results.db> [out.csv]   prints per-kernel totals (whole run) and an analysis of the
LAST train step (delimited by the Adam launch): launches, span, summed kernel time, gaps."""
import collections
import csv
import re
import sqlite3
import sys

import numpy as np

db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name, start, end from kernels order by start"))


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"\((?:[^()]|\([^()]*\))*\)\s*$", "", n)       # drop the argument list
    n = re.sub(r"\(GemmParams\)|\(StepArgs.*$", "", n)
    return n[:110]


agg = collections.defaultdict(list)
for n, s, e in rows:
    agg[short(n)].append(e - s)
tot = sum(sum(v) for v in agg.values())
table = sorted(agg.items(), key=lambda kv: -sum(kv[1]))
if len(sys.argv) > 2:
    with open(sys.argv[2], "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
        for k, v in table:
            a = np.array(v, dtype=np.float64)
            w.writerow([k, len(v), int(a.sum()), f"{a.mean():.1f}", f"{100 * a.sum() / tot:.2f}", int(a.min()),
                        int(a.max()), f"{a.std():.1f}"])
idx = [i for i, r in enumerate(rows) if "adam_dev" in r[0] or "adam_kernel" in r[0]]
if len(idx) >= 2:
    step = rows[idx[-2] + 1: idx[-1] + 1]
    span = (step[-1][2] - step[0][1]) / 1e6
    busy = sum(e - s for _, s, e in step) / 1e6
    print(f"last step: {len(step)} launches, span {span:.3f} ms, summed kernel time {busy:.3f} ms")
    st = collections.defaultdict(lambda: [0, 0])
    for n, s, e in step:
        st[short(n)][0] += 1
        st[short(n)][1] += e - s
    for k, v in sorted(st.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[3]) if len(sys.argv) > 3 else 30]:
        print(f"{k[:90]:90s} {v[0]:5d} {v[1] / 1e6:8.3f} ms  avg {v[1] / v[0] / 1e3:8.2f} us")
    gaps = [(step[i + 1][1] - step[i][2], short(step[i][0])[:40], short(step[i + 1][0])[:40]) for i in range(len(step) - 1)]
    gaps.sort(reverse=True)
    print("largest gaps (us):", [(round(g / 1e3, 1), a, b) for g, a, b in gaps[:6]])
