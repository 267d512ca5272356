#!/bin/bash
# usage: tools/pmc_ta.sh <tag> [postings] [lib]  -- texture-addresser / L1 (TCP) passes: is the per-CU
# vector-memory front end the bottleneck of the decode kernel?
TAG=${1:-pta}; N=${2:-4e8}; LIB=$3
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
[ -n "$LIB" ] && export DINT_HIP_LIB=$LIB
pass() { n=$1; shift; timeout 150 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$n -- python3 $R/tools/quick_bench.py $N 8192 > $OUT/$n.log 2>&1; }
# at most two TA/TD and four TCP counters fit one pass ("exceeds the capabilities of the hardware" otherwise)
pass ta1 TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum
pass ta2 TA_BUFFER_TOTAL_CYCLES_sum TA_BUFFER_WAVEFRONTS_sum
pass ta3 TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum
pass ta4 TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum
pass ta5 TD_TD_BUSY_sum TD_TC_STALL_sum
pass ta6 GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_VMEM
python3 - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$OUT/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: [0.0, 0])
    for row in csv.DictReader(open(f)):
        if "decode" not in row["Kernel_Name"]: continue
        a = agg[row["Counter_Name"]]; a[0] += float(row["Counter_Value"]); a[1] += 1
    for k, (v, n) in agg.items(): print(f"{k:40s} per-launch {v / n:.5g}")
PY
grep -il "error\|invalid\|not found" $OUT/*.log | head
