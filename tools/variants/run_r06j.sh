R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r06j; mkdir -p $OUT; V=$R/dint_amd/variants
cd $R
timeout 500 python3 tools/ab_bench.py --type multi_packed_dint --unit-ints 256 --table --postings 1e9 --rounds 4 --reps 3 h6=$V/h6.so te2=$V/te2.so ts=$V/ts.so cw1=$V/cw1.so > $OUT/ab_multi_ts.txt 2>&1; tail -5 $OUT/ab_multi_ts.txt
MULTI="--type multi_packed_dint --unit-ints 256"
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py $MULTI --steps 5 --warmup 2 --cpu-seconds 0 --no-verify"
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- $B > $OUT/fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- $B > $OUT/write.log 2>&1
cd $R
python3 tools/pmc_traffic.py $OUT/fetch $OUT/write multi_packed_dint 5000000000 $OUT/traffic_multi.json decode_multi | tee $OUT/traffic_multi.log
timeout 600 python3 bench.py $MULTI --steps 20 --warmup 5 --traffic-file $OUT/traffic_multi.json > $OUT/bench_multi.json 2> $OUT/bench_multi.err; cut -c1-400 $OUT/bench_multi.json
