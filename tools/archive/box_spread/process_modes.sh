#!/bin/bash
# usage: tools/box_spread/process_modes.sh <tag> [processes] — the same bench command in N fresh processes on ONE box, each under
# one counter set: does the kernel's time differ from process to process, and what moves with it?
TAG=${1:-modes}; N=${2:-8}
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --steps 3 --warmup 1 --cpu-seconds 0 --no-verify"
SETS=("SQC_ICACHE_REQ SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_IFETCH_LEVEL"
      "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_TAG_STALL_sum TCC_REQ_sum")
for i in $(seq 1 $N); do
  set=${SETS[$(( (i - 1) % 2 ))]}
  timeout 400 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -- $B > $OUT/p$i.log 2>&1
done
python3 - <<PY
import csv, glob, collections
for i in range(1, $N + 1):
    dur = []
    for f in glob.glob(f"$OUT/p{i}/**/*kernel_trace.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if "decode_single_kernel" in row["Kernel_Name"]:
                dur.append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e6)
    agg = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(f"$OUT/p{i}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if "decode_single_kernel" not in row["Kernel_Name"]: continue
            a = agg[row["Counter_Name"]]; a[0] += float(row["Counter_Value"]); a[1] += 1
    line = f"process {i}: kernel ms " + " ".join(f"{d:.3f}" for d in dur) + " | " + "  ".join(f"{k} {v[0] / max(1, v[1]):.4g}" for k, v in sorted(agg.items()))
    print(line)
PY
