#!/bin/bash
# GPU box: bench.py with the clocks and the socket power sampled while it runs (rocm-smi every 0.3 s): does a slow run
# have slower clocks, a lower power cap? usage: tools/clock_watch.sh [bench args]
rocm-smi --showmaxpower 2>/dev/null | grep -i "power" | sed 's/GPU\[0\]\s*: //'
( while true; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "fclk|mclk|sclk|socclk|Power" | sed 's/GPU\[0\]\s*: //' | sed 's/clock level: [0-9]*: //' | tr '\n' ' '; echo; sleep 0.3; done ) > /tmp/clocks.log &
W=$!
python3 $GRAFT_REPO_ROOT/bench.py --cpu-seconds 0 --steps 50 "$@" 2> /tmp/bench.err | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('RESULT kernel_ms', r['kernel_ms'], r['kernel_ms_min_median_max'], 'shader_mhz', r['shader_mhz'], r.get('box_probe'))"
kill $W
# the samples taken under load (the timed loop: 400 launches, 3 s)
grep -E "Power \(W\): [5-9][0-9][0-9]|Power \(W\): 1[0-9][0-9][0-9]" /tmp/clocks.log | sed 's/=* Power Consumption =*//; s/Current Socket Graphics Package //; s/socclk clock level: S: //' | head -n 14
