"""Average of every collected counter per kernel name from rocprofv3 --pmc CSVs: pmc_by_kernel.py <dir>"""
import collections, csv, glob, re, sys
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = re.sub(r"\(anonymous namespace\)::|^void ", "", r["Kernel_Name"])
        n = re.sub(r"\(.*$", "", n)[:60]
        acc[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
for n, cs in sorted(acc.items()):
    print(n, {c: (len(v), round(sum(v) / len(v), 1)) for c, v in cs.items()})
