R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r06f; mkdir -p $OUT; V=$R/dint_amd/variants
cd $R
timeout 500 python3 tools/ab_bench.py --type multi_packed_dint --unit-ints 256 --table --postings 1e9 --rounds 4 --reps 3 ra1=$V/ra1.so pf1=$V/pf1.so > $OUT/ab_multi_pf.txt 2>&1; tail -3 $OUT/ab_multi_pf.txt
for v in ra1 pf1; do DINT_HIP_LIB=$V/$v.so PLACEMENT_TRIALS=2 timeout 300 python3 tools/inindex_bench.py 1e8 $OUT/inindex_$v.json > $OUT/inindex_$v.log 2>&1; done
python3 - <<PY
import json
for f in ("ra1","pf1"):
    try:
        d=json.load(open("$OUT/inindex_%s.json"%f))
        for t in ("single_packed_dint","multi_packed_dint"):
            print(f,t,d[t]["docs_and_freqs"]["ms"],d[t]["docs_only"]["ms"],d[t]["bit_exact"])
    except Exception as e: print(f,"failed",e)
PY
