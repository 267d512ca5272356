R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r06h; mkdir -p $OUT; V=$R/dint_amd/variants
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_index.py tests/test_gpu_fuzz.py tests/test_gpu_queries.py -m gpu -x -q > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
timeout 500 python3 tools/ab_bench.py --type multi_packed_dint --unit-ints 256 --table --postings 1e9 --rounds 4 --reps 3 h6=$V/h6.so te2=$V/te2.so > $OUT/ab_multi_te.txt 2>&1; tail -3 $OUT/ab_multi_te.txt
for v in h6 te2; do DINT_HIP_LIB=$V/$v.so PLACEMENT_TRIALS=2 timeout 300 python3 tools/inindex_bench.py 1e8 $OUT/inindex_$v.json > $OUT/inindex_$v.log 2>&1; done
python3 - <<PY
import json
for f in ("h6","te2"):
    try:
        d=json.load(open("$OUT/inindex_%s.json"%f))
        for t in ("single_packed_dint","multi_packed_dint"):
            print(f,t,d[t]["docs_and_freqs"]["ms"],d[t]["docs_only"]["ms"],d[t]["bit_exact"])
    except Exception as e: print(f,"failed",e)
PY
timeout 900 python3 tools/inindex_scale.py $OUT/inindex_5e9_single.json 5e9 single_packed_dint > $OUT/inindex_5e9_single.log 2>&1; tail -2 $OUT/inindex_5e9_single.log | cut -c1-1500
