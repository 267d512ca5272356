R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r06k; mkdir -p $OUT; V=$R/dint_amd/variants
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_bench.py -m gpu -x -q > $OUT/pytest.log 2>&1; tail -5 $OUT/pytest.log
timeout 500 python3 tools/ab_bench.py --type multi_packed_dint --unit-ints 256 --table --postings 1e9 --rounds 4 --reps 3 te2=$V/te2.so sp1=$V/sp1.so > $OUT/ab_multi_sp.txt 2>&1; tail -3 $OUT/ab_multi_sp.txt
