R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r06d; mkdir -p $OUT; V=$R/dint_amd/variants
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_index.py tests/test_gpu_fuzz.py tests/test_gpu_queries.py -m gpu -x -q > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
timeout 500 python3 tools/ab_bench.py --type multi_packed_dint --unit-ints 256 --table --postings 1e9 --rounds 4 --reps 3 cw0=$V/cw0.so cw1=$V/cw1.so > $OUT/ab_multi_cw.txt 2>&1; tail -3 $OUT/ab_multi_cw.txt
DINT_HIP_LIB=$V/cw0.so PLACEMENT_TRIALS=2 timeout 300 python3 tools/inindex_bench.py 1e8 $OUT/inindex_cw0.json > $OUT/inindex_cw0.log 2>&1
DINT_HIP_LIB=$V/cw1.so PLACEMENT_TRIALS=2 timeout 300 python3 tools/inindex_bench.py 1e8 $OUT/inindex_cw1.json > $OUT/inindex_cw1.log 2>&1
python3 - <<PY
import json
for f in ("cw0","cw1"):
    try:
        d=json.load(open("$OUT/inindex_%s.json"%f))
        for t in ("single_packed_dint","multi_packed_dint"):
            print(f,t,d[t]["docs_and_freqs"]["ms"],d[t]["docs_only"]["ms"],d[t]["bit_exact"])
    except Exception as e: print(f,"failed",e)
PY
