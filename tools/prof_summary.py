#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace --stats CSV directory (kernel_stats.csv)."""
import csv
import glob
import sys

d = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
f = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"# {f}\n# total kernel time {tot / 1e6:.2f} ms")
for r in rows[:n]:
    print(f"{r['Name'][:88]:88s} n={r['Calls']:>5s} tot={float(r['TotalDurationNs']) / 1e6:8.2f}ms "
          f"avg={float(r['AverageNs']) / 1e3:8.1f}us {float(r['Percentage']):5.1f}%")
