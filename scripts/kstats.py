"""Summarise a rocprofv3 kernel_stats.csv: python scripts/kstats.py <csv> <n_steps_profiled>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 18]:
    print(f"{r['Name'][:78]:78s} calls/step={float(r['Calls'])/n:7.1f} ms/step={float(r['TotalDurationNs'])/1e6/n:8.3f} avg_us={float(r['AverageNs'])/1e3:8.2f}")
print("kernel ms/step", tot / 1e6 / n)
