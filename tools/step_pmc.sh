#!/bin/bash
# Per-kernel averages of arbitrary PMC counters over a bench workload's training steps (one rocprofv3 --pmc pass per counter
# GROUP, kernel-trace only: groups are separated by "+"; counters of a group share a pass).
# usage (GPU box, repo root): tools/step_pmc.sh <outdir> <workload> "CNT_A CNT_B+CNT_C"   -> <outdir>/pmc_per_kernel.json
OUT=${1:-gpurun_out/pmc}; WL=${2:-swinir_x8}; GROUPS_=${3:-"TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum"}
ROOT=$(pwd)
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
IFS='+' read -ra GRP <<< "$GROUPS_"
for g in "${GRP[@]}"; do
  rocprofv3 --kernel-trace --pmc $g --output-format csv -d "$ROOT/$OUT/pass$i" -- \
    python3 "$ROOT/bench.py" --workload $WL --steps 3 --warmup 2 --train-only --no-roofline > "$ROOT/$OUT/bench$i.log" 2>&1 || true
  i=$((i+1))
done
cd "$ROOT"
python3 - "$OUT" <<'P'
import csv, glob, json, re, sys, collections
d = sys.argv[1]
out = collections.defaultdict(dict)
for pdir in sorted(glob.glob(f"{d}/pass*")):
    cc = glob.glob(f"{pdir}/**/*counter_collection.csv", recursive=True)
    kt = glob.glob(f"{pdir}/**/*kernel_trace.csv", recursive=True)
    if not cc or not kt:
        continue
    dur = {r["Dispatch_Id"]: int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(kt[0]))}
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    tdur = collections.defaultdict(lambda: [0.0, 0])
    seen = set()
    for r in csv.DictReader(open(cc[0])):
        m = re.search(r"k_\w+(<[^>]*>)?", r["Kernel_Name"])
        k = m.group(0) if m else r["Kernel_Name"][:40]
        a = acc[k][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
        if r["Dispatch_Id"] not in seen and r["Dispatch_Id"] in dur:
            seen.add(r["Dispatch_Id"]); tdur[k][0] += dur[r["Dispatch_Id"]]; tdur[k][1] += 1
    for k, cs in acc.items():
        for c, (v, n) in cs.items():
            out[k][c] = v / max(n, 1)
        if tdur[k][1]:
            out[k].setdefault("launches", tdur[k][1]); out[k].setdefault("avg_us_under_pmc", tdur[k][0] / tdur[k][1] / 1e3)
json.dump(out, open(f"{d}/pmc_per_kernel.json", "w"), indent=1)
rows = sorted(out.items(), key=lambda kv: -kv[1].get("avg_us_under_pmc", 0) * kv[1].get("launches", 0))[:14]
for k, v in rows:
    print(f"{k[:44]:44s} n={v.get('launches', 0):4d} {v.get('avg_us_under_pmc', 0):8.1f} us  " +
          "  ".join(f"{c}={x:.3g}" for c, x in v.items() if c not in ("launches", "avg_us_under_pmc")))
P
rm -rf "$OUT"/pass*
