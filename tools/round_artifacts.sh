#!/bin/bash
# GPU box: everything a round's profiles/ is made of, after tools/profile_round.sh (the headline's artefacts): config 3
# (multi_packed_dint, block-granular units) with its kernel stats and SQ counters, the in-index decode, the query timings,
# and the headline on the process's FIRST allocation in three consecutive fresh processes.
# usage: tools/round_artifacts.sh r04   -> gpurun_out/<tag>x/
TAG=${1:-r04}; R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/${TAG}x; mkdir -p $OUT
cd $R
for i in 1 2 3; do
  timeout 400 python3 bench.py --placement-trials 1 --cpu-seconds 0 --steps 20 --warmup 5 > $OUT/first_alloc_$i.json 2> $OUT/first_alloc_$i.err
done
python3 - <<PY | tee $OUT/first_allocation_x3.txt
import json
for i in (1, 2, 3):
    d = json.load(open("$OUT/first_alloc_%d.json" % i))
    print("fresh process %d, --placement-trials 1: %.1f G ints/s, roofline.frac %.4f, kernel ms min/median/max %s, bit_exact %s"
          % (i, d["value"] / 1e3, d["roofline"]["frac"], d["roofline"]["kernel_ms_min_median_max"], d["bit_exact"]))
PY
MULTI="--type multi_packed_dint --unit-ints 256"
timeout 600 python3 bench.py $MULTI --steps 20 --warmup 5 > $OUT/bench_multi.json 2> $OUT/bench_multi.err; cat $OUT/bench_multi.json | cut -c1-300
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py $MULTI --steps 5 --warmup 2 --cpu-seconds 0 --no-verify"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_multi -- $B > $OUT/stats_multi.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/sq1 -- $B > $OUT/sq1.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- $B > $OUT/fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- $B > $OUT/write.log 2>&1
cd $R
INTS=$(python3 -c "import json; print(json.load(open('$OUT/bench_multi.json'))['config']['ints_per_gpu_per_step'])")
python3 tools/pmc_traffic.py $OUT/fetch $OUT/write multi_packed_dint $INTS $OUT/traffic_multi.json decode_multi | tee $OUT/traffic_multi.log
cp $OUT/bench_multi.json $OUT/bench.json; python3 tools/pmc_sq_summary.py $OUT $INTS | tee $OUT/sq_multi.txt
find $OUT/stats_multi -name "*kernel_stats.csv" -exec cat {} \; | head -8 | tee $OUT/kernel_stats_multi_head.csv
timeout 300 python3 tools/inindex_bench.py > $OUT/inindex.json 2> $OUT/inindex.err; cat $OUT/inindex.json | cut -c1-300
timeout 900 python3 tests/query_timing.py --forms > $OUT/queries_1e8.json 2> $OUT/queries.err; cat $OUT/queries_1e8.json | cut -c1-600
