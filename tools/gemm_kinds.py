#!/usr/bin/env python3
"""Per-kind durations of the NT GEMM launches of the SwinIR training step from a rocprofv3 kernel trace
(rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 bench.py ...):  python tools/gemm_kinds.py DIR
A step issues 192 k_ntw (k_ntp with SRHIP_NTW=0) launches: forward 24 x (qkv, proj, fc1, fc2), backward 24 x (fc2 dgrad, fc1 dgrad,
proj dgrad, qkv dgrad)."""
import collections, csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "k_ntp" in r["Kernel_Name"] or "k_ntw" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
kf = ["qkv fwd  N540 K180 LN-prologue", "proj fwd N180 K180 +residual +row stats", "fc1 fwd  N360 K180 LN-prologue",
      "fc2 fwd  N180 K360 GELU-prologue +residual +row stats"]
kb = ["fc2 dgrad N360 K180 *gelu'(h), gelu(h) out", "fc1 dgrad N180 K360 LayerNorm-backward epilogue",
      "proj dgrad N180 K180 DropPath scale", "qkv dgrad N180 K540 LayerNorm-backward epilogue"]
d = collections.defaultdict(list)
for i, r in enumerate(rows):
    j = i % 192
    d[kf[j % 4] if j < 96 else kb[(j - 96) % 4]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = 0.0
for k in kf + kb:
    v = sorted(d[k]); tot += sum(v) / len(v)
    print(f"{k:56s} n={len(v):4d}  avg {sum(v)/len(v):6.1f} us  min {v[0]:6.1f}")
print(f"sum per Swin block {tot:.0f} us  (x 24 blocks = {tot * 24 / 1e3:.2f} ms per step)")
