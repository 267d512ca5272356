R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r06i; mkdir -p $OUT
cd $R
timeout 1200 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; tail -3 $OUT/pytest_gpu.log
timeout 1800 python3 tools/inindex_scale.py $OUT/inindex_scale.json 1e8,1e9,5e9 single_packed_dint,multi_packed_dint > $OUT/inindex_scale.log 2>&1
python3 - <<PY
import json
d=json.load(open("$OUT/inindex_scale.json"))
for r in d["rows"]:
    print(r["type"], r["postings"], "ready", r["ready"], "qitems", r["table"]["docs_queue_items"], r["table"]["freqs_queue_items"], "d+f", r["docs_and_freqs"]["ms"], r["docs_and_freqs"]["frac_of_8TBps"], "docs", r["docs_only"]["ms"], r["docs_only"]["frac_of_8TBps"], r["bit_exact"])
print(json.dumps(d.get("fit")))
PY
