#!/bin/bash
export MASP_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1
for q in 8 12 16; do
  MASTER_PORT=2956$q GPU_MAX_HW_QUEUES=$q MASP_BENCH_E2E=0 python bench.py --steps 8 --warmup 2 --no-cpu-baseline 2> /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('dist queues $q', round(d['value'],1), round(d['resident']['value'],1))"
done
unset MASP_BENCH_FORCE_DIST
MASP_BENCH_E2E=0 python bench.py --steps 8 --warmup 2 --no-cpu-baseline 2> /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('plain', round(d['value'],1), round(d['resident']['value'],1))"
