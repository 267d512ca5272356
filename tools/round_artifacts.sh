#!/bin/bash
# GPU box: everything a round's profiles/ is made of, after tools/profile_round.sh (the headline's artefacts): config 3
# (multi_packed_dint, block-granular units) with its kernel stats and SQ counters, the in-index decode and its launches side by
# side (rocprofv3 kernel trace -> tools/kernel_timeline.py), the query timings.
# usage: tools/round_artifacts.sh r04   -> gpurun_out/<tag>x/
TAG=${1:-r04}; R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/${TAG}x; mkdir -p $OUT
cd $R
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
timeout 300 python3 tools/inindex_bench.py 1e8 $OUT/inindex.json > $OUT/inindex.log 2> $OUT/inindex.err; cut -c1-300 $OUT/inindex.log
(cd /tmp && PLACEMENT_TRIALS=1 timeout 400 rocprofv3 --kernel-trace --output-format csv -d $OUT/inindex_trace -- python3 $R/tools/inindex_bench.py > $OUT/inindex_trace.log 2>&1)
python3 tools/kernel_timeline.py $OUT/inindex_trace --calls 400 --gap-us 40 --match decode_single_index,interpolative,finalize_flagged,fillBuffer > $OUT/inindex_timeline_all.txt 2>&1
timeout 900 python3 tests/query_timing.py --forms > $OUT/queries_1e8.json 2> $OUT/queries.err; cat $OUT/queries_1e8.json | cut -c1-600
