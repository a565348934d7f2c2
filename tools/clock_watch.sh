#!/bin/bash
# sclk / power / temperature sampled twice a second while the default bench runs (is the chip clock-limited under this load?)
out=${1:-gpurun_out/clock_watch.txt}
( while true; do echo "$(date +%s.%N | cut -c1-14) $(rocm-smi --showclocks --showpower --showtemp --csv 2>/dev/null | tail -n +2 | head -1)"; sleep 0.5; done ) > $out.samples &
W=$!
rocm-smi --showclocks --showpower --csv 2>/dev/null | head -1 > $out
MASP_BENCH_E2E=0 python bench.py --steps 24 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench', d['value'], d['resident']['value'])" >> $out
kill $W
cat $out.samples >> $out; rm -f $out.samples
