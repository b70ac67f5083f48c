#!/bin/bash
# GPU box: the rocprofv3 evidence of one round, written under gpurun_out/prof_<tag>/ and summarised into profiles/.
#   bash scripts/profile_round.sh <tag>        e.g. r2a
# 1. --kernel-trace --stats of the default bench command (graph replay): per-kernel totals + the last step's launches;
# 2. three --pmc passes (FETCH_SIZE | WRITE_SIZE | MFMA busy) over eager steps (counters need per-dispatch records),
#    each in its own run, with --kernel-trace only (no other trace domain next to --pmc);
# 3. pmc_traffic.json (to be copied to profiles/) for bench.py's roofline.traffic, keyed by the hash of the kernel sources.
set -u
TAG="${1:-r2}"
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/prof_$TAG
SUM=$OUT/summary
mkdir -p "$OUT" "$SUM"
BENCH="bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs --no-roofline"
rocprofv3 --kernel-trace --stats -f csv rocpd -d "$OUT/kt" -o kt -- python3 $BENCH > "$OUT/kt.json" 2> "$OUT/kt.err"
DB=$(find "$OUT/kt" -name '*results.db' | head -1)
CSV=$(find "$OUT/kt" -name '*kernel_stats.csv' | head -1)
if [ -n "$DB" ]; then python3 scripts/rocpd_stats.py "$DB" "$SUM/${TAG}_kernel_stats.csv" 40 > "$SUM/${TAG}_last_step.txt" 2>&1;
elif [ -n "$CSV" ]; then cp "$CSV" "$SUM/${TAG}_kernel_stats.csv"; fi
tail -1 "$OUT/kt.json" > "$SUM/${TAG}_bench_under_profiler.json"
# 1b. the same for BASELINE configs[2] (bf16, B=128, T=256)
BENCH16="bench.py --dtype bf16 --batch 128 --frames 256 --steps 6 --warmup 3 --no-cpu-baseline --no-other-configs --no-roofline"
rocprofv3 --kernel-trace --stats -f csv rocpd -d "$OUT/kt16" -o kt16 -- python3 $BENCH16 > "$OUT/kt16.json" 2> "$OUT/kt16.err"
DB16=$(find "$OUT/kt16" -name '*results.db' | head -1)
if [ -n "$DB16" ]; then python3 scripts/rocpd_stats.py "$DB16" "$SUM/${TAG}_bf16_kernel_stats.csv" 40 > "$SUM/${TAG}_bf16_last_step.txt" 2>&1; fi
tail -1 "$OUT/kt16.json" > "$SUM/${TAG}_bf16_bench_under_profiler.json"
# 1c. ... and for the per-GPU shape of configs[4] (bf16, B=64, T=512)
BENCH4="bench.py --dtype bf16 --batch 64 --frames 512 --steps 6 --warmup 3 --no-cpu-baseline --no-other-configs --no-roofline"
rocprofv3 --kernel-trace --stats -f csv rocpd -d "$OUT/kt4" -o kt4 -- python3 $BENCH4 > "$OUT/kt4.json" 2> "$OUT/kt4.err"
DB4=$(find "$OUT/kt4" -name '*results.db' | head -1)
if [ -n "$DB4" ]; then python3 scripts/rocpd_stats.py "$DB4" "$SUM/${TAG}_bf16_c4_kernel_stats.csv" 40 > "$SUM/${TAG}_bf16_c4_last_step.txt" 2>&1; fi
tail -1 "$OUT/kt4.json" > "$SUM/${TAG}_bf16_c4_bench_under_profiler.json"
# algorithmic bytes per instantiation (bench.py's own tags), for the fabric / algorithmic column of the PMC summary
python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs > "$OUT/alg.json" 2> "$OUT/alg.err"
EAGER="bench.py --steps 2 --warmup 1 --graph 0 --no-cpu-baseline --no-other-configs --no-roofline"
for C in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  D="$OUT/pmc_$(echo $C | tr ' ' '_')"
  rocprofv3 --pmc $C --kernel-trace -f csv -d "$D" -o pmc -- python3 $EAGER > "$D.json" 2> "$D.err"
done
python3 scripts/pmc_summary.py "$OUT" 5 "$TAG" "$SUM/pmc_traffic.json" > "$SUM/${TAG}_pmc_summary.md" 2> "$OUT/pmc_summary.err"
ls -la "$SUM"
# gpurun merges gpurun_out/ back: copy $SUM/* into profiles/ (tracked) afterwards
