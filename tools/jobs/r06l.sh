#!/bin/bash
# round 6: 4 slots against 3 at the default 16 hardware queues (round 5 measured 4 slots at 24 queues, where the process walks into queue oversubscription)
o=gpurun_out/r06l; mkdir -p $o
for rep in 1 2 3; do
  for cfg in "3 3" "4 4" "3 4"; do
    set -- $cfg
    v=$(MASP_HIP_SLOTS=$1 MASP_BENCH_H2H_CALLS=$2 GPU_MAX_HW_QUEUES=16 MASP_BENCH_OTHER=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('value %.1f resident %.1f e2e %.1f (first %.1f) lone %.2f sclk %s W %s valu %.3f' % (d['value'], d['resident']['value'], d['end_to_end']['value'], d['end_to_end']['first_call']['value'], d['single_proof_latency_ms'], d['clocks']['value_region']['sclk_mhz_mean'], d['clocks']['value_region']['socket_power_w_mean'], d['roofline_valu']['frac']))")
    echo "slots $1 calls in flight $2: $v" | tee -a $o/slots_3_4_at_16_queues.txt
  done
done
