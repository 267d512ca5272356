#!/bin/bash
# GPU box: the judged artefacts of a round — GPU tests, bench.py (default command), rocprofv3 kernel stats and
# the FETCH_SIZE / WRITE_SIZE passes of the same bench command, then the bench line again with the measured
# traffic attached. usage: tools/profile_round.sh r02 [extra bench args]
TAG=${1:-r02}; shift
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; tail -3 $OUT/pytest_gpu.log
python3 bench.py "$@" > $OUT/bench.json 2> $OUT/bench.err; cat $OUT/bench.json
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $R/bench.py --steps 5 --warmup 2 --cpu-seconds 0 --no-verify $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $BENCH > $OUT/stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- $BENCH > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- $BENCH > $OUT/write.log 2>&1
cd $R
INTS=$(python3 -c "import json; print(json.load(open('$OUT/bench.json'))['config']['ints_per_gpu_per_step'])")
TYPE=$(python3 -c "import json; print(json.load(open('$OUT/bench.json'))['metric'].split('vroom ')[1].rstrip(')'))")
python3 tools/pmc_traffic.py $OUT/fetch $OUT/write $TYPE $INTS $OUT/traffic.json | tee $OUT/traffic.log
python3 bench.py --cpu-seconds 0 --traffic-file $OUT/traffic.json "$@" > $OUT/bench_with_traffic.json 2>> $OUT/bench.err; cat $OUT/bench_with_traffic.json
find $OUT/stats -name "*kernel_stats.csv" -exec cat {} \; | head -6 | tee $OUT/kernel_stats_head.csv
