#!/bin/bash
# GPU box: the judged artefacts of a round — GPU tests, bench.py (default command), rocprofv3 kernel stats, the
# FETCH_SIZE / WRITE_SIZE passes and one SQ pass of the same bench command, then the bench line again with the
# measured traffic attached. Every profiler pass runs under its own `timeout` (a pass that aborts can hang until the box's limit).
# usage: [SKIP_TESTS=1] tools/profile_round.sh r04 [extra bench args]
TAG=${1:-r03}; shift
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
if [ -z "$SKIP_TESTS" ]; then python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; tail -3 $OUT/pytest_gpu.log; fi
python3 bench.py "$@" > $OUT/bench.json 2> $OUT/bench.err; cat $OUT/bench.json
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $R/bench.py --steps 5 --warmup 2 --cpu-seconds 0 --no-verify $*"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $BENCH > $OUT/stats.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- $BENCH > $OUT/fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- $BENCH > $OUT/write.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/sq1 -- $BENCH > $OUT/sq1.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_FLAT SQ_ACTIVE_INST_SCA --output-format csv -d $OUT/sq2 -- $BENCH > $OUT/sq2.log 2>&1
cd $R
INTS=$(python3 -c "import json; print(json.load(open('$OUT/bench.json'))['config']['ints_per_gpu_per_step'])")
TYPE=$(python3 -c "import json; print(json.load(open('$OUT/bench.json'))['metric'].split('vroom ')[1].rstrip(')'))")
python3 tools/pmc_traffic.py $OUT/fetch $OUT/write $TYPE $INTS $OUT/traffic.json | tee $OUT/traffic.log
python3 tools/pmc_sq_summary.py $OUT $INTS | tee $OUT/sq.txt
python3 bench.py --cpu-seconds 0 --traffic-file $OUT/traffic.json "$@" > $OUT/bench_with_traffic.json 2>> $OUT/bench.err; cat $OUT/bench_with_traffic.json
find $OUT/stats -name "*kernel_stats.csv" -exec cat {} \; | head -8 | tee $OUT/kernel_stats_head.csv
# (the stats average over EVERY launch of the decode kernel in the profiled process — bench.py's placement trials during
# set-up included, some of them into buffers the bench then drops; the timed region's launches are the last ones)
python3 - <<PY | tee $OUT/kernel_stats_timed.txt
import csv, glob
rows = []
for f in glob.glob("$OUT/stats/**/*kernel_trace.csv", recursive=True):
    rows += [r for r in csv.DictReader(open(f)) if "decode_" in r["Kernel_Name"] and "_kernel" in r["Kernel_Name"] and "query" not in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows]
timed = dur[-5:]
print(f"decode kernel launches in the profiled process: {len(dur)}; all: mean {sum(dur) / max(1, len(dur)):.4f} ms; the timed region's last 5: "
      + " ".join(f"{d:.4f}" for d in timed) + f" ms, mean {sum(timed) / max(1, len(timed)):.4f} ms")
PY
