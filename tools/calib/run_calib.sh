#!/bin/bash
# on the GPU box, from the repo root: tools/calib/run_calib.sh -> gpurun_out/mfma_calib/summary.txt
set -e
ROOT=$(pwd); OUT=$ROOT/gpurun_out/mfma_calib; rm -rf $OUT; mkdir -p $OUT
hipcc --offload-arch=gfx950 -O2 tools/calib/mfma_busy_calib.hip -o /tmp/mfma_busy_calib
/tmp/mfma_busy_calib > $OUT/timing.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $OUT/pmc -- /tmp/mfma_busy_calib > /dev/null 2>&1
cd $ROOT
python3 - $OUT <<'P'
import csv, glob, sys, collections
d = sys.argv[1]
kt = glob.glob(f"{d}/pmc/**/*kernel_trace.csv", recursive=True)[0]
cc = glob.glob(f"{d}/pmc/**/*counter_collection.csv", recursive=True)[0]
dur = {r["Dispatch_Id"]: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(kt))}
acc = collections.defaultdict(list)
for r in csv.DictReader(open(cc)):
    if r["Counter_Name"] == "SQ_VALU_MFMA_BUSY_CYCLES" and r["Dispatch_Id"] in dur:
        ns, name = dur[r["Dispatch_Id"]]
        acc["16x16x32" if "16x16x32" in name else "32x32x16"].append((float(r["Counter_Value"]), ns))
with open(f"{d}/summary.txt", "w") as f:
    f.write(open(f"{d}/timing.txt").read())
    for k, v in acc.items():
        c, ns = v[-1]
        n = (80000 if k == "16x16x32" else 40000) * 1024
        f.write(f"{k}: counter {c:.4g} over {ns / 1e3:.1f} us; per MFMA {c / n:.2f} counts; busy fraction at 2.4 GHz x 1024 SIMDs "
                f"{c / (ns * 2.4 * 1024):.3f}\n")
print(open(f"{d}/summary.txt").read())
P
