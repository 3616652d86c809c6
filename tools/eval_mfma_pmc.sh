#!/bin/bash
# Matrix-core busy share of the kernels of one network's evaluation forward (tools/mb_eval_net.py), as tools/step_mfma_pmc.sh does for
# a training workload.  usage (GPU box, repo root): tools/eval_mfma_pmc.sh <outdir> <net_type> <method> [amp]
OUT=${1:-gpurun_out/eval_mfma}; NET=${2:-VDSR}; METHOD=${3:-VDSR}; AMP=${4:-1}
ROOT=$(pwd)
mkdir -p "$OUT"; rm -rf "$OUT/sq_$NET"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d "$ROOT/$OUT/sq_$NET" -- \
  python3 "$ROOT/tools/mb_eval_net.py" $NET $METHOD 8 2 $AMP > "$ROOT/$OUT/$NET.log" 2>&1 || true
cd "$ROOT"
python3 - "$OUT" "$NET" "$AMP" <<'P'
import csv, glob, json, re, sys, collections
d, net, amp = sys.argv[1], sys.argv[2], sys.argv[3]
cc = glob.glob(f"{d}/sq_{net}/**/*counter_collection.csv", recursive=True)
kt = glob.glob(f"{d}/sq_{net}/**/*kernel_trace.csv", recursive=True)
dur = {r["Dispatch_Id"]: int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(kt[0]))}
acc = collections.defaultdict(lambda: [0.0, 0.0, 0])
for r in csv.DictReader(open(cc[0])):
    if r["Counter_Name"] != "SQ_VALU_MFMA_BUSY_CYCLES" or r["Dispatch_Id"] not in dur:
        continue
    m = re.search(r"k_\w+(<[^>]*>)?", r["Kernel_Name"])
    k = m.group(0) if m else r["Kernel_Name"][:40]
    a = acc[k]; a[0] += float(r["Counter_Value"]); a[1] += dur[r["Dispatch_Id"]]; a[2] += 1
out = {k: {"launches": n, "avg_us_under_pmc": t / n / 1e3, "mfma_busy_frac_at_2.4GHz": c / (t * 1e-9 * 2.4e9 * 1024.0)}
       for k, (c, t, n) in acc.items() if c > 0}
path = f"{d}/eval_mfma_busy.json"
try:
    allo = json.load(open(path))
except Exception:
    allo = {}
allo[f"{net} x8 amp={amp}"] = out
json.dump(allo, open(path, "w"), indent=1)
for k, v in sorted(out.items(), key=lambda kv: -kv[1]["avg_us_under_pmc"] * kv[1]["launches"])[:4]:
    print(f"{net:8s} {k[:44]:44s} n={v['launches']:5d} avg {v['avg_us_under_pmc']:8.1f} us  MFMA-busy >= {100 * v['mfma_busy_frac_at_2.4GHz']:.1f} %")
P
rm -rf "$OUT/sq_$NET"
