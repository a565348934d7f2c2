#!/bin/bash
# the two modes of a bench process (2.5 % apart) against the number of hardware queues: 15 slot streams share 8 by default
for r in 1 2 3 4 5; do for q in 8 16 24; do
  GPU_MAX_HW_QUEUES=$q MASP_BENCH_E2E=0 python bench.py --steps 8 --warmup 2 --no-cpu-baseline 2> /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('queues $q: %.1f  resident %.1f  (%.1f ms per step)' % (d['value'], d['resident']['value'], d['resident']['ms_per_step'] if 'ms_per_step' in d['resident'] else d['ms_per_step']))"
done; done
