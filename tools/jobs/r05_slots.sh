#!/bin/bash
# 3 slots / 16 queues (default) against 4 slots / 24 queues, bench.py with the driver's flags, alternating
for i in 1 2; do
  for cfg in "3 16" "4 24"; do
    set -- $cfg
    MASP_HIP_SLOTS=$1 GPU_MAX_HW_QUEUES=$2 MASP_BENCH_OTHER=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); e=d['end_to_end']
print('slots $1 queues $2: value %.1f resident %.1f e2e %.1f (first %.1f) lone %.2f valu_frac %.3f sclk %.0f' % (d['value'], d['resident']['value'], e['value'], e['first_call']['value'], d['single_proof_latency_ms'], d['roofline_valu']['frac'], d['roofline_valu']['sclk_mhz']))"
  done
done
