R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r06t; mkdir -p $OUT; V=$R/dint_amd/variants
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_abi.py tests/test_gpu_bench.py -m gpu -x -q > $OUT/pytest.log 2>&1; tail -5 $OUT/pytest.log
timeout 500 python3 tools/ab_bench.py --type multi_packed_dint --unit-ints 16384 --table --postings 1e9 --rounds 4 --reps 3 wf=$V/wf.so rf=$V/rf.so > $OUT/ab_multi16k_refine.txt 2>&1; tail -4 $OUT/ab_multi16k_refine.txt
timeout 500 python3 tools/ab_bench.py --type multi_packed_dint --unit-ints 65536 --table --postings 1e9 --rounds 3 --reps 3 wf=$V/wf.so rf=$V/rf.so > $OUT/ab_multi64k_refine.txt 2>&1; tail -4 $OUT/ab_multi64k_refine.txt
