#!/bin/bash
# usage: tools/box_spread/pair_counters.sh <tag> — pair_counters.py under five L2 / memory counter sets; the last 8 decode
# launches of each process alternate slow, fast
TAG=${1:-pairs}; R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" \
           "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum" \
           "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum" \
           "TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_LEVEL_sum TCC_NORMAL_EVICT_sum TCC_NORMAL_WRITEBACK_sum" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -- python3 $R/tools/box_spread/pair_counters.py > $OUT/p$i.log 2>&1
  tail -5 $OUT/p$i.log | head -2
done
python3 - <<PY
import csv, glob, collections
for i in range(1, 6):
    rows = []
    for f in glob.glob(f"$OUT/p{i}/**/*counter_collection.csv", recursive=True):
        rows += [r for r in csv.DictReader(open(f)) if "decode_single_kernel" in r["Kernel_Name"]]
    ids = sorted({int(r["Dispatch_Id"]) for r in rows})[-8:]
    for name in sorted({r["Counter_Name"] for r in rows}):
        v = {int(r["Dispatch_Id"]): float(r["Counter_Value"]) for r in rows if r["Counter_Name"] == name}
        s = [v[k] for k in ids[0::2] if k in v]; f_ = [v[k] for k in ids[1::2] if k in v]
        if s and f_:
            print(f"{name:36s} slow {sum(s) / len(s):.5g}   fast {sum(f_) / len(f_):.5g}   slow/fast {sum(s) / len(s) / max(1e-9, sum(f_) / len(f_)):.3f}")
PY
