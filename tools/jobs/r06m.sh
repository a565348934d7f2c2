#!/bin/bash
# round 6: slots 4 / 5 at 16 hardware queues (and 4 with smaller tree sub-batches); first / second / third context of a process at 4 slots / 16 queues
o=gpurun_out/r06m; mkdir -p $o
for rep in 1 2; do
  for cfg in "4 86" "5 86" "4 64" "5 64"; do
    set -- $cfg
    v=$(MASP_HIP_SLOTS=$1 MASP_HIP_TREE_SUB=$2 GPU_MAX_HW_QUEUES=16 MASP_BENCH_OTHER=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('value %.1f resident %.1f e2e %.1f lone %.2f sclk %s W %s valu %.3f' % (d['value'], d['resident']['value'], d['end_to_end']['value'], d['single_proof_latency_ms'], d['clocks']['value_region']['sclk_mhz_mean'], d['clocks']['value_region']['socket_power_w_mean'], d['roofline_valu']['frac']))")
    echo "slots $1 tree sub-batch $2: $v" | tee -a $o/slots_4_5_at_16_queues.txt
  done
done
echo "=== MASP_HIP_SLOTS=4 GPU_MAX_HW_QUEUES=16" | tee -a $o/contexts_at_4_slots_16_queues.txt
MASP_HIP_SLOTS=4 GPU_MAX_HW_QUEUES=16 timeout 700 python tools/second_context_stage_probe.py 2>&1 | grep "context\|own streams\|host to host\|isolated" | tee -a $o/contexts_at_4_slots_16_queues.txt
