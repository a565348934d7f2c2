#!/bin/bash
o=gpurun_out/r04h; mkdir -p $o
for i in 1 2; do
  for mode in plain dist; do
    if [ $mode = dist ]; then export MASP_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=2955$i; else unset MASP_BENCH_FORCE_DIST; fi
    MASP_BENCH_E2E=0 python bench.py --steps 8 --warmup 2 --no-cpu-baseline 2> $o/$mode$i.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$mode', round(d['value'],1), round(d['resident']['value'],1), d['rccl_ranks'], d.get('collectives'))"
  done
done
