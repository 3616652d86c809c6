#!/usr/bin/env python3
"""GRBM_GUI_ACTIVE per dispatch + kernel-trace durations -> effective clock per kernel (GHz) = counter / 8 / duration."""
import csv
import glob
import json
import sys

d, out = sys.argv[1], sys.argv[2]
cc = glob.glob(f"{d}/**/*counter_collection.csv", recursive=True)[0]
kt = glob.glob(f"{d}/**/*kernel_trace.csv", recursive=True)[0]
dur = {}
for r in csv.DictReader(open(kt)):
    dur[r["Dispatch_Id"]] = (r["Kernel_Name"], int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
acc = {}
for r in csv.DictReader(open(cc)):
    if r["Counter_Name"] != "GRBM_GUI_ACTIVE":
        continue
    k = dur.get(r["Dispatch_Id"])
    if k is None or k[1] <= 0:
        continue
    a = acc.setdefault(k[0], [0.0, 0.0, 0])
    a[0] += float(r["Counter_Value"]); a[1] += k[1]; a[2] += 1
res = {}
tot_c = tot_t = 0.0
for k, (c, t, n) in acc.items():
    res[k] = {"launches": n, "avg_us": t / n / 1e3, "effective_clock_ghz": c / 8.0 / t}
    tot_c += c; tot_t += t
res["__all_kernels__"] = {"effective_clock_ghz": tot_c / 8.0 / tot_t, "kernel_time_ms": tot_t / 1e6}
json.dump(res, open(out, "w"), indent=1)
for k, v in sorted(res.items(), key=lambda kv: -kv[1].get("avg_us", 0) * kv[1].get("launches", 0))[:12]:
    print(f"{k[:72]:72s} {v.get('launches', 0):5d} x {v.get('avg_us', 0):8.1f} us  {v['effective_clock_ghz']:.2f} GHz")
print("all kernels:", res["__all_kernels__"])
