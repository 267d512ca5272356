#!/bin/bash
# usage: tools/pmc_l2.sh <tag> <bench args...>  -- L2 hit/miss and memory-request counters of the decode kernel
TAG=$1; shift
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --steps 3 --warmup 1 --cpu-seconds 0 --no-verify $*"
timeout 300 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum --output-format csv -d $OUT/a -- $B > $OUT/a.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc TCC_EA_RDREQ_sum TCC_EA_WRREQ_sum TCC_WRITE_sum TCC_EA_RDREQ_32B_sum --output-format csv -d $OUT/b -- $B > $OUT/b.log 2>&1
python3 - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$OUT/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: [0.0, 0])
    for row in csv.DictReader(open(f)):
        if "decode_single" not in row["Kernel_Name"]: continue
        a = agg[row["Counter_Name"]]; a[0] += float(row["Counter_Value"]); a[1] += 1
    for k, (v, n) in agg.items(): print(f"{k:28s} per-launch {v / n:.5g}")
PY
