#!/bin/bash
# usage: tools/pmc_l2.sh <tag> [lib.so] -- L2 hit/miss, L1<->L2 request latencies and memory-request stalls of the decode kernel
TAG=$1; LIB=$2
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
[ -n "$LIB" ] && export DINT_HIP_LIB=$LIB
B="python3 $R/bench.py --steps 3 --warmup 1 --cpu-seconds 0 --no-verify --postings 1e9 --replicate 2"
i=0
for set in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" \
           "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum" \
           "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum" \
           "TCC_TAG_STALL_sum TCC_BUSY_sum TCC_CYCLE_sum TCC_STREAMING_REQ_sum" \
           "TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_LEVEL_sum TCC_NORMAL_EVICT_sum TCC_NORMAL_WRITEBACK_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -- $B > $OUT/p$i.log 2>&1
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: [0.0, 0])
for f in sorted(glob.glob("$OUT/**/*counter_collection.csv", recursive=True)):
    for row in csv.DictReader(open(f)):
        if "decode_single" not in row["Kernel_Name"] and "decode_multi" not in row["Kernel_Name"]: continue
        a = agg[row["Counter_Name"]]; a[0] += float(row["Counter_Value"]); a[1] += 1
with open("$OUT/l2.txt", "w") as o:
    for k in sorted(agg):
        v, n = agg[k]
        line = f"{k:36s} per-launch {v / n:.5g}"
        print(line); o.write(line + "\n")
PY
