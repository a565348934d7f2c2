#!/bin/bash
# round 6, ninth GPU call: the build with a slot's side streams created by its first lone proof (a batch-only prover: one stream per slot) —
# lone / golden / hardware-queue tests, then bench.py at GPU_MAX_HW_QUEUES 8 / 12 / 16 alternating (driver flags, no cpu baseline)
o=gpurun_out/r06i; mkdir -p $o
timeout 900 python -m pytest tests/test_golden_proofs.py tests/test_gpu_lone_and_warm.py tests/test_gpu_hw_queues.py tests/test_gpu_lone_graph.py tests/test_gpu_capi_harness.py tests/test_capi_harness.py -m gpu -x -q > $o/tests.txt 2>&1; tail -3 $o/tests.txt
for rep in 1 2; do
  for q in 8 12 16; do
    v=$(GPU_MAX_HW_QUEUES=$q python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); o=d['other_circuits']; print('value %.1f resident %.1f e2e %.1f lone %.2f output %.0f convert %.0f stage %.2f hwq %s' % (d['value'], d['resident']['value'], d['end_to_end']['value'], d['single_proof_latency_ms'], o['output']['value'], o['convert']['value'], d['roofline']['avg_launch_ms'], d['config']['hw_queues']))")
    echo "GPU_MAX_HW_QUEUES=$q: $v" | tee -a $o/hw_queues_8_12_16_ab.txt
  done
done
