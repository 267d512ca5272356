R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r06o; mkdir -p $OUT; V=$R/dint_amd/variants
cd $R
timeout 500 python3 tools/ab_bench.py --type multi_packed_dint --unit-ints 256 --table --postings 1e9 --rounds 5 --reps 3 base=$V/base.so wf=$V/wf.so > $OUT/ab_multi_wf.txt 2>&1; tail -4 $OUT/ab_multi_wf.txt
timeout 500 python3 tools/ab_bench.py --type multi_packed_dint --unit-ints 16384 --postings 1e9 --rounds 4 --reps 3 base=$V/base.so wf=$V/wf.so > $OUT/ab_multi16k_wf.txt 2>&1; tail -4 $OUT/ab_multi16k_wf.txt
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_index.py tests/test_gpu_queries.py -m gpu -x -q > $OUT/pytest.log 2>&1; tail -5 $OUT/pytest.log
