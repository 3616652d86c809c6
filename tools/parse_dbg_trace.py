#!/usr/bin/env python3
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "k_ntb" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
variants = [0, 8, 4, 12, 16, 32, 48, 56, 58, 62, 2]
nshape = 6
assert len(d) == len(variants) * nshape * 5, len(d)
print("dbg bits: 2 no LDS stores | 4 no MFMA | 8 no epilogue | 16 no W loads | 32 no A loads   (GPU durations, us, median of 5)")
i = 0
for v in variants:
    row = []
    for s in range(nshape):
        x = sorted(d[i:i + 5]); i += 5
        row.append(x[2])
    print(f"dbg={v:3d}: " + "  ".join(f"{t:6.1f}" for t in row))
