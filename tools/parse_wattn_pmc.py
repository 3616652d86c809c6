#!/usr/bin/env python3
"""SQ / GRBM counters of the window-attention launches -> JSON (profiles/rNN_wattn_pmc.json).
mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (kernel cycles x SIMDs), kernel cycles = GRBM_GUI_ACTIVE / 8
(the counter is summed over the 8 XCDs, MI355X_MICROARCH.md "DVFS give-back"), 1024 SIMDs on the chip.
Also the same share computed from the launch duration of the kernel trace at the measured clock."""
import csv
import glob
import json
import re
import sys

d = sys.argv[1]


def table(sub, pattern):
    f = glob.glob(f"{d}/{sub}/**/*{pattern}", recursive=True)
    return list(csv.DictReader(open(f[0]))) if f else []


acc = {}
for r in table("sq", "counter_collection.csv") + table("grbm", "counter_collection.csv"):
    k = r["Kernel_Name"]
    if "wattn" not in k:
        continue
    e = acc.setdefault(re.search(r"k_wattn_\w+", k).group(0), {})
    c = e.setdefault(r["Counter_Name"], [0.0, 0])
    c[0] += float(r["Counter_Value"])
    c[1] += 1
dur = {}
for r in table("trace", "kernel_stats.csv"):
    if "wattn" in r["Name"]:
        dur[re.search(r"k_wattn_\w+", r["Name"]).group(0)] = float(r["AverageNs"]) / 1e3
out = {}
for k, e in acc.items():
    avg = {c: v[0] / v[1] for c, v in e.items()}
    cyc = avg.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
    o = {"counters_per_launch": avg, "avg_launch_us": dur.get(k)}
    if cyc and "SQ_VALU_MFMA_BUSY_CYCLES" in avg:
        o["kernel_cycles"] = cyc
        o["mfma_busy_frac"] = avg["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024.0)
        if dur.get(k):
            o["clock_ghz"] = cyc / (dur[k] * 1e3)
    if dur.get(k) and "SQ_VALU_MFMA_BUSY_CYCLES" in avg:      # lower bound: the chip clocks below 2.4 GHz under load
        o["mfma_busy_frac_at_2.4GHz"] = avg["SQ_VALU_MFMA_BUSY_CYCLES"] / (dur[k] * 1e-6 * 2.4e9 * 1024.0)
    if "SQ_BUSY_CYCLES" in avg and "SQ_VALU_MFMA_BUSY_CYCLES" in avg:
        o["mfma_busy_over_sq_busy"] = avg["SQ_VALU_MFMA_BUSY_CYCLES"] / avg["SQ_BUSY_CYCLES"]
    out[k] = o
print(json.dumps(out, indent=1))
