#!/bin/bash
for r in 1 2 3 4 5 6 7; do for q in 8 16; do
  GPU_MAX_HW_QUEUES=$q MASP_BENCH_E2E=0 python bench.py --steps 8 --warmup 2 --no-cpu-baseline 2> /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('queues $q: %.1f  resident %.1f  lat %.2f' % (d['value'], d['resident']['value'], d['single_proof_latency_ms']))"
done; done
