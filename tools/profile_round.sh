#!/bin/bash
# GPU box: run the judged artefacts of a round: GPU tests, bench.py, rocprofv3 kernel stats and the
# FETCH_SIZE / WRITE_SIZE passes of the same bench command. usage: tools/profile_round.sh r01
TAG=${1:-r01}
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; tail -3 $OUT/pytest_gpu.log
python bench.py > $OUT/bench.json 2> $OUT/bench.err; cat $OUT/bench.json
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $R/bench.py --steps 5 --warmup 1 --cpu-seconds 0"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $BENCH > $OUT/stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- $BENCH > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- $BENCH > $OUT/write.log 2>&1
cd $R
python3 tools/pmc_traffic.py $OUT/fetch $OUT/write single_packed_dint 1e9 | tee $OUT/traffic.log
cp profiles/traffic.json $OUT/traffic.json
find $OUT/stats -name "*kernel_stats.csv" -exec cat {} \; | head -5
