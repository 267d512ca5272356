#!/bin/bash
# usage: tools/pmc_tlb.sh <tag> [runs]  -- UTCL1 translation counters of the decode kernel, several processes
TAG=${1:-tlb}; RUNS=${2:-4}
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for i in $(seq 1 $RUNS); do
  timeout 300 rocprofv3 --kernel-trace --pmc TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_PERMISSION_MISS_sum --output-format csv -d $OUT/r$i -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-seconds 0 --no-verify > $OUT/r$i.log 2>&1
  python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: [0.0, 0]); dur=[]
for f in glob.glob("$OUT/r$i/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "decode_single" not in row["Kernel_Name"]: continue
        a = agg[row["Counter_Name"]]; a[0] += float(row["Counter_Value"]); a[1] += 1
for f in glob.glob("$OUT/r$i/**/*kernel_trace.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "decode_single" in row["Kernel_Name"]: dur.append((int(row["End_Timestamp"])-int(row["Start_Timestamp"]))/1e6)
print("run $i kernel ms", [round(d,3) for d in dur[-4:]], {k: round(v/n) for k,(v,n) in agg.items()})
PY
done
