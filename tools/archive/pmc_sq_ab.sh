#!/bin/bash
# usage: tools/pmc_sq_ab.sh <tag> name=lib.so [name=lib.so ...] -- the SQ counter sets of several builds, side by side
TAG=$1; shift
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --steps 3 --warmup 1 --cpu-seconds 0 --no-verify --postings 1e9 --replicate 2"
for spec in "$@"; do
  name=${spec%%=*}; lib=${spec#*=}
  export DINT_HIP_LIB=$R/$lib
  i=0
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU" \
             "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA" \
             "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS SQ_VMEM_TA_ADDR_FIFO_FULL"; do
    i=$((i+1))
    timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/$name.p$i -- $B > $OUT/$name.p$i.log 2>&1
  done
done
python3 - "$@" <<PY
import csv, glob, collections, sys
names = [s.split("=")[0] for s in sys.argv[1:]]
table = collections.defaultdict(dict)
for name in names:
    agg = collections.defaultdict(lambda: [0.0, 0])
    for f in sorted(glob.glob(f"$OUT/{name}.p*/**/*counter_collection.csv", recursive=True)):
        for row in csv.DictReader(open(f)):
            if "decode_single" not in row["Kernel_Name"] and "decode_multi" not in row["Kernel_Name"]: continue
            a = agg[row["Counter_Name"]]; a[0] += float(row["Counter_Value"]); a[1] += 1
    for k, (v, n) in agg.items(): table[k][name] = v / n
with open("$OUT/sq_ab.txt", "w") as o:
    hdr = f"{'counter (per 2e9-integer launch, /int)':40s}" + "".join(f"{n:>16s}" for n in names)
    print(hdr); o.write(hdr + "\n")
    for k in sorted(table):
        line = f"{k:40s}" + "".join(f"{table[k].get(n, float('nan')) / 2e9:16.4f}" for n in names)
        print(line); o.write(line + "\n")
PY
