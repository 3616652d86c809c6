#!/bin/bash
# Matrix-core busy share of EVERY kernel of a bench workload: SQ_VALU_MFMA_BUSY_CYCLES per launch against launch duration x 2.4 GHz
# x 1024 SIMDs (a lower bound: the chip clocks lower under load).  SQ counters in their own pass, kernel-trace only.
# usage (GPU box, repo root): tools/step_mfma_pmc.sh <outdir> [workload]   -> <outdir>/mfma_busy_per_kernel.json
OUT=${1:-gpurun_out/mfma_pmc}; WL=${2:-swinir_x8}
ROOT=$(pwd)
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export SRHIP_SWIN_SIDE_WGRAD=0     # launches in order on one stream: per-kernel counters are attributable (see refresh_profiles.sh)
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d "$ROOT/$OUT/sq" -- \
  python3 "$ROOT/bench.py" --workload $WL --steps 4 --warmup 2 --train-only --no-roofline > "$ROOT/$OUT/bench.log" 2>&1 || true
cd "$ROOT"
python3 - "$OUT" <<'P'
import csv, glob, json, re, sys, collections
d = sys.argv[1]
cc = glob.glob(f"{d}/sq/**/*counter_collection.csv", recursive=True)
kt = glob.glob(f"{d}/sq/**/*kernel_trace.csv", recursive=True)
dur = {}
for r in csv.DictReader(open(kt[0])):
    dur[r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
acc = collections.defaultdict(lambda: [0.0, 0.0, 0])
for r in csv.DictReader(open(cc[0])):
    if r["Counter_Name"] != "SQ_VALU_MFMA_BUSY_CYCLES" or r["Dispatch_Id"] not in dur:
        continue
    m = re.search(r"k_\w+(<[^>]*>)?", r["Kernel_Name"])
    k = m.group(0) if m else r["Kernel_Name"][:40]
    a = acc[k]; a[0] += float(r["Counter_Value"]); a[1] += dur[r["Dispatch_Id"]]; a[2] += 1
out = {}
for k, (c, t, n) in acc.items():
    if c <= 0: continue
    out[k] = {"launches": n, "avg_us_under_pmc": t / n / 1e3, "mfma_busy_cycles_per_launch": c / n,
              "mfma_busy_frac_at_2.4GHz": c / (t * 1e-9 * 2.4e9 * 1024.0)}
sys.path.insert(0, "sr-caco-2_amd")
try:                      # tie the table to the kernel sources it measured (bench.py reads it only when they match the loaded build)
    import importlib.util
    spec = importlib.util.spec_from_file_location("probe_hash", "sr-caco-2_amd/srhip/probe.py")
    src = open("sr-caco-2_amd/srhip/probe.py").read()
    import hashlib, os
    h = hashlib.sha256()
    cd = "sr-caco-2_amd/csrc"
    for f in sorted(os.listdir(cd)):
        if f.endswith((".hip", ".h")) or f == "Makefile":
            h.update(f.encode()); h.update(open(os.path.join(cd, f), "rb").read())
    meta = {"csrc_sha16": h.hexdigest()[:16]}
except Exception as e:
    meta = {"csrc_sha16": None, "error": str(e)}
json.dump(dict(out, _meta=meta), open(f"{d}/mfma_busy_per_kernel.json", "w"), indent=1)
for k, v in sorted(out.items(), key=lambda kv: -kv[1]["avg_us_under_pmc"] * kv[1]["launches"])[:12]:
    print(f"{k[:50]:50s} n={v['launches']:5d} avg {v['avg_us_under_pmc']:8.1f} us  MFMA-busy >= {100 * v['mfma_busy_frac_at_2.4GHz']:.1f} %")
P
rm -rf "$OUT/sq"
