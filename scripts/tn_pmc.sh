#!/bin/bash
# GPU box: fabric traffic and L2 hit rate of the row-contiguous (weight-gradient) form, tall kernel against the 256 x 256 kernel
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
for K in 0 2; do
  for C in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
    D=gpurun_out/tnpmc_${K}_$(echo $C | tr ' ' '_')
    DVAE_GEMM_256=$K rocprofv3 --pmc $C --kernel-trace -f csv -d "$D" -o p -- python3 scripts/tn_probe.py 10 > "$D.out" 2> "$D.err"
    F=$(find "$D" -name '*counter_collection.csv' | head -1)
    echo "== DVAE_GEMM_256=$K $C"
    python3 - "$F" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"][:60]
    if "gemm" not in k: continue
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
for k, d in agg.items():
    print("  ", k, {c: f"{v / cnt[(k, c)]:.4g} per launch" for c, v in d.items()})
PY
  done
done
