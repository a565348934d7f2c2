#!/bin/bash
# a live torch + RCCL runtime (one-rank process group) against the plain run, at the new default of 16 hardware queues and at 8
sel='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("%.1f  resident %.1f" % (d["value"], d["resident"]["value"]))'
for r in 1 2 3; do
  for q in 16 8; do
    echo "plain, $q queues: $(GPU_MAX_HW_QUEUES=$q MASP_BENCH_E2E=0 python bench.py --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "$sel")"
    echo "one-rank RCCL, $q queues: $(GPU_MAX_HW_QUEUES=$q MASP_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=2957$r MASP_BENCH_E2E=0 python bench.py --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "$sel")"
  done
done
