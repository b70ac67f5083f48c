"""Aggregate rocprofv3 --pmc counter_collection CSVs per kernel family.
usage: pmc_summary.py <dir with pass sub-dirs> <n_steps>   (prints a markdown table)"""
import collections, csv, glob, re, sys
root, steps = sys.argv[1], float(sys.argv[2])
fam = lambda n: ("gemm_f32_kernel" if "gemm_f32" in n else "lstm_step_* (H>=512)" if "lstm_step" in n else
                 "lstm_seq_*_h64" if "lstm_seq" in n else "adam" if "adam" in n else "bn_*" if "bn_" in n else
                 "colsum" if "colsum" in n else "other")
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = fam(r["Kernel_Name"])
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[k][r["Counter_Name"]] += 1
print("| kernel family | launches/step | MFMA busy (SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 x 1024 SIMDs)) | HBM-side read MB/step (2 x FETCH_SIZE KB) | write MB/step (WRITE_SIZE KB) |")
print("|---|---|---|---|---|")
for k in sorted(acc, key=lambda k: -acc[k].get("GRBM_GUI_ACTIVE", 0)):
    a = acc[k]
    n = cnt[k].get("GRBM_GUI_ACTIVE", 0) / steps
    busy = a.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / max(1.0, a.get("GRBM_GUI_ACTIVE", 0) / 8 * 1024)
    print(f"| {k} | {n:.0f} | {busy:.2f} | {2 * a.get('FETCH_SIZE', 0) / 1024 / steps:.0f} | {a.get('WRITE_SIZE', 0) / 1024 / steps:.0f} |")
