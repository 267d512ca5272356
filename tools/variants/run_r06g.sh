R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r06g; mkdir -p $OUT
cd $R
for wl in gov2-bpi59 gov2-exceptions gov2-freqs; do
  timeout 600 python3 bench.py --workload $wl --steps 20 --warmup 5 --cpu-seconds 8 > $OUT/bench_$wl.json 2> $OUT/bench_$wl.err
done
timeout 600 python3 bench.py --type single_rect_dint --steps 20 --warmup 5 --cpu-seconds 8 > $OUT/bench_rect.json 2> $OUT/bench_rect.err
timeout 600 python3 bench.py --workload clueweb --steps 20 --warmup 5 --cpu-seconds 8 > $OUT/bench_clueweb.json 2> $OUT/bench_clueweb.err
timeout 300 python3 tools/index_stream_rate.py 5e8 $OUT/index_stream_rate.json > $OUT/index_stream_rate.log 2>&1
timeout 1500 python3 tools/inindex_scale.py $OUT/inindex_scale.json 1e8,1e9,5e9 single_packed_dint,multi_packed_dint > $OUT/inindex_scale.log 2>&1
timeout 1200 python3 tools/emulate_ranks.py --world 8 --out $OUT/config4_emulated.json -- --workload clueweb --type single_packed_dint --steps 10 --warmup 3 --cpu-seconds 0 > $OUT/config4.log 2>&1
python3 - <<PY
import json,glob
for f in sorted(glob.glob("$OUT/bench_*.json")):
    try:
        d=json.load(open(f)); c=d["config"]
        print(f.split("/")[-1], d["value"], "frac", d["roofline"]["frac"], "first", d["roofline"]["frac_first_allocation"], "bpi", c["bits_per_int"], "ipc", c["ints_per_codeword"], "exc", c["exception_pct"], "lds", c["lds_hit_pct"], "cpu", d["cpu_baseline"]["value"] if d["cpu_baseline"] else None, d["bit_exact"])
    except Exception as e: print(f, "failed", e)
PY
tail -3 $OUT/index_stream_rate.log; tail -2 $OUT/inindex_scale.log | cut -c1-900; tail -2 $OUT/config4.log | cut -c1-600
