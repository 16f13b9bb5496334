#!/bin/bash
# sample power / clocks with rocm-smi while a command runs: tools/smi_watch.sh <outfile> <cmd...>
out=$1; shift
( while true; do rocm-smi --showpower --showclocks --showtemp --json 2>/dev/null | python3 -c "
import sys, json, time
try:
    d = json.load(sys.stdin)['card0']
    keys = [k for k in d if any(s in k.lower() for s in ('power', 'sclk', 'mclk', 'fclk', 'junction', 'hotspot', 'edge'))]
    print(round(time.time(), 2), {k: d[k] for k in keys})
except Exception as e:
    print('err', e)
"; sleep 0.15; done ) > $out 2>&1 &
W=$!
"$@"
kill $W
