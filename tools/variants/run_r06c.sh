R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r06c; mkdir -p $OUT
cd $R
timeout 600 python -m pytest tests/test_gpu_queries.py tests/test_gpu_index.py tests/test_gpu_bench.py -m gpu -x -q > $OUT/pytest_new.log 2>&1; tail -3 $OUT/pytest_new.log
timeout 700 python3 bench.py --placement-trials 6 --probe-eval > $OUT/bench_lottery_probe_eval.json 2> $OUT/bench_lottery.err; tail -c 600 $OUT/bench_lottery.err | tail -3
timeout 500 python3 bench.py --placement-trials 1 --place-by-probe --cpu-seconds 0 > $OUT/bench_by_probe_1.json 2> $OUT/bench_by_probe_1.err
timeout 500 python3 bench.py --placement-trials 1 --place-by-probe --cpu-seconds 0 > $OUT/bench_by_probe_2.json 2> $OUT/bench_by_probe_2.err
python3 - <<PY
import json
for f in ("bench_lottery_probe_eval","bench_by_probe_1","bench_by_probe_2"):
    try:
        d=json.load(open("$OUT/"+f+".json"))
        print(f, d["value"], d["roofline"]["frac"], d["roofline"]["frac_first_allocation"], d["config"].get("placement_probe"), d["config"].get("placement_probe_eval"))
    except Exception as e: print(f, "failed", e)
PY
