#!/bin/bash
# GPU box: per-kernel totals of the default bench command under rocprofv3 (the first pass of profile_round.sh only)
set -u
TAG="${1:-q}"
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/kt_$TAG
mkdir -p "$OUT"
rocprofv3 --kernel-trace --stats -f csv rocpd -d "$OUT/kt" -o kt -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs --no-roofline > "$OUT/kt.json" 2> "$OUT/kt.err"
DB=$(find "$OUT/kt" -name '*results.db' | head -1)
python3 scripts/rocpd_stats.py "$DB" "$OUT/kernel_stats.csv" 60 > "$OUT/last_step.txt" 2>&1
rm -rf "$OUT/kt"
