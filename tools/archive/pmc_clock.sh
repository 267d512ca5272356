#!/bin/bash
# usage: tools/pmc_clock.sh <tag> [runs] -- effective shader clock of the decode kernel (GRBM_GUI_ACTIVE / duration), several processes
TAG=${1:-clk}; RUNS=${2:-4}
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for i in $(seq 1 $RUNS); do
  timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUT/r$i -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-seconds 0 --no-verify > $OUT/r$i.log 2>&1
  python3 - <<PY
import csv, glob
cyc={}; dur={}
for f in glob.glob("$OUT/r$i/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "decode_single" in row["Kernel_Name"]: cyc[row["Dispatch_Id"]] = float(row["Counter_Value"])
for f in glob.glob("$OUT/r$i/**/*kernel_trace.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "decode_single" in row["Kernel_Name"]: dur[row["Dispatch_Id"]] = (int(row["End_Timestamp"])-int(row["Start_Timestamp"]))
out=[]
for k in sorted(cyc, key=int)[-4:]:
    if k in dur: out.append((round(dur[k]/1e6,3), round(cyc[k]/8/dur[k],3)))
print("run $i (kernel ms, GHz):", out)
PY
done
