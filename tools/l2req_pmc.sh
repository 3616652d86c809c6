#!/bin/bash
# L1 -> L2 read requests per launch of the NT GEMM microbenchmark (tools/mb_ntw.py arm): how many bytes a launch pulls
# through the CUs' L1s.  usage (GPU box, repo root): tools/l2req_pmc.sh <outdir>
OUT=${1:-gpurun_out/l2req}
ROOT=$(pwd)
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$ROOT/$OUT/run" -- \
  python3 "$ROOT/tools/mb_ntw.py" arm > "$ROOT/$OUT/mb.log" 2>&1 || { tail -5 "$ROOT/$OUT/mb.log"; rocprofv3 --list-avail 2>/dev/null | grep -o "TCP_[A-Z_]*READ[A-Z_]*\|TCC_[A-Z_]*REQ[A-Z_]*" | sort -u | head -40; }
cd "$ROOT"
python3 - "$OUT" <<'P'
import csv, glob, sys, collections
d = sys.argv[1]
cc = glob.glob(f"{d}/run/**/*counter_collection.csv", recursive=True)
kt = glob.glob(f"{d}/run/**/*kernel_trace.csv", recursive=True)
if not cc or not kt:
    print("no counter output"); sys.exit(0)
name = {}
for r in csv.DictReader(open(kt[0])):
    name[r["Dispatch_Id"]] = (r["Kernel_Name"], r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Grid_Size_Y", ""))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(cc[0])):
    k = name.get(r["Dispatch_Id"])
    if k is None or "k_nt" not in k[0]:
        continue
    acc[(k[0][:60], k[1], k[2])][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print(k, {c: (len(x), sum(x) / len(x)) for c, x in v.items()})
P
rm -rf "$OUT/run"
