#!/bin/bash
# GPU box: bench.py with the clocks sampled while it runs (rocm-smi every 0.5 s): does a slow run have slower clocks?
# usage: tools/clock_watch.sh [bench args]
( while true; do rocm-smi --showclocks 2>/dev/null | grep -E "fclk|mclk|sclk|socclk" | sed 's/GPU\[0\]\s*: //' | tr '\n' ' '; echo; sleep 0.5; done ) > /tmp/clocks.log &
W=$!
python3 $GRAFT_REPO_ROOT/bench.py --cpu-seconds 0 "$@" 2> /tmp/bench.err | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('RESULT kernel_ms', r['kernel_ms'], 'shader_mhz', r['shader_mhz'])"
kill $W
grep "device buffers" /tmp/bench.err
# the samples taken while the timed loop ran: the last ones before the bench ended
tail -n 6 /tmp/clocks.log | sort | uniq -c
