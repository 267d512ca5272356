#!/bin/bash
# GPU box: latency / stall counters of the decode kernel (separate --pmc passes over a short bench run).
# usage: tools/pmc_latency.sh <tag> [bench args]
TAG=${1:-lat}; shift
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $R/bench.py --steps 4 --warmup 2 --cpu-seconds 0 --no-verify --postings 1e9 --replicate 2 $*"
i=0
for set in "LdsLatency VmemLatency" \
           "SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES" \
           "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_INSTS_LDS_ATOMIC SQ_INSTS_LDS_LOAD_BANDWIDTH SQ_INSTS_LDS_STORE_BANDWIDTH SQ_INSTS_LDS_ATOMIC_BANDWIDTH SQ_INSTS_BRANCH SQ_INSTS_VALU"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -- $BENCH > $OUT/p$i.log 2>&1
done
cd $R
python3 - <<PY
import collections, csv, glob
agg = collections.defaultdict(lambda: [0.0, 0])
for f in sorted(glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True)):
    for row in csv.DictReader(open(f)):
        if not row["Kernel_Name"].startswith(("dint_dev::decode_single_kernel", "dint_dev::decode_multi_kernel")):
            continue
        a = agg[row["Counter_Name"]]; a[0] += float(row["Counter_Value"]); a[1] += 1
with open("$OUT/latency.txt", "w") as o:
    for k in sorted(agg):
        v, n = agg[k]
        line = f"{k:32s} per launch {v / n:14.6g}   ({n} launches)"
        print(line); o.write(line + "\n")
PY
