R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r06n; mkdir -p $OUT; V=$R/dint_amd/variants
cd $R
timeout 600 python3 tools/ab_bench.py --type single_packed_dint --unit-ints 16384 --postings 1e9 --rounds 5 --reps 3 base=$V/base.so t896=$V/t896.so t768=$V/t768.so t640=$V/t640.so > $OUT/ab_single_threads.txt 2>&1; tail -8 $OUT/ab_single_threads.txt
timeout 500 python3 tools/ab_bench.py --type multi_packed_dint --unit-ints 256 --table --postings 1e9 --rounds 4 --reps 3 base=$V/base.so t896=$V/t896.so t768=$V/t768.so > $OUT/ab_multi_threads.txt 2>&1; tail -6 $OUT/ab_multi_threads.txt
