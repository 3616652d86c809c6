#!/usr/bin/env python3
"""FETCH_SIZE / WRITE_SIZE per launch for each kernel -> JSON (profiles/).
gfx950 corrections (MI355X_MICROARCH.md, HBM): counters are in KiB; FETCH_SIZE
reports half the bytes of a wide coalesced read -> doubled; WRITE_SIZE is exact
for 16-B/lane streaming stores (our GEMM epilogue stores 4 B/lane: uncalibrated,
reported as read)."""
import csv
import glob
import json
import sys

d, out = sys.argv[1], sys.argv[2]
res = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"{d}/{c}/**/*counter_collection.csv", recursive=True)[0]
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != c:
            continue
        k = r["Kernel_Name"]
        e = res.setdefault(k, {"FETCH_SIZE": [0.0, 0], "WRITE_SIZE": [0.0, 0]})
        e[c][0] += float(r["Counter_Value"])
        e[c][1] += 1
summary = {}
for k, e in res.items():
    if e["FETCH_SIZE"][1] == 0 or e["WRITE_SIZE"][1] == 0:
        continue
    fetch = e["FETCH_SIZE"][0] / e["FETCH_SIZE"][1] * 1024 * 2
    write = e["WRITE_SIZE"][0] / e["WRITE_SIZE"][1] * 1024
    summary[k] = {"launches": e["FETCH_SIZE"][1], "fetch_bytes_per_launch_x2": fetch,
                  "write_bytes_per_launch": write, "hbm_bytes_per_launch": fetch + write}
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "sr-caco-2_amd"))
try:                     # which build these counters belong to (bench.py uses a profile only for the same kernel sources)
    from srhip.probe import csrc_hash
    meta = {"csrc_sha16": csrc_hash()}
except Exception as e:   # noqa: BLE001
    meta = {"csrc_sha16": None, "error": str(e)}
json.dump(dict(summary, _meta=meta), open(out, "w"), indent=1)
for k, v in sorted(summary.items(), key=lambda t: -t[1]["hbm_bytes_per_launch"] * t[1]["launches"])[:12]:
    print(f"{k[:70]:70s} n={v['launches']:5d} fetch*2={v['fetch_bytes_per_launch_x2']/1e6:8.1f}MB "
          f"write={v['write_bytes_per_launch']/1e6:8.1f}MB")
