#!/bin/bash
# round 6: stream creation order 3 (slot 0's side streams, all main streams, the other side streams) against order 1 (all mains first): lone latency in bench.py and contexts A / B / C
o=gpurun_out/r06p; mkdir -p $o
for rep in 1 2 3; do
  for L in tools/_build/ab/libmasp_hip_order3.so tools/_build/ab/libmasp_hip_order1.so; do
    v=$(MASP_HIP_LIBRARY=$PWD/$L MASP_BENCH_E2E=0 MASP_BENCH_OTHER=0 python bench.py --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('value %.1f resident %.1f lone %.2f / resident witness %.2f' % (d['value'], d['resident']['value'], d['single_proof_latency_ms'], d['single_proof_latency']['resident_witness_ms']))")
    echo "$L: $v" | tee -a $o/stream_order_ab.txt
  done
done
for L in tools/_build/ab/libmasp_hip_order3.so; do
  echo "=== $L" | tee -a $o/contexts.txt
  MASP_HIP_LIBRARY=$PWD/$L timeout 700 python tools/second_context_stage_probe.py 2>&1 | grep "context\|own streams\|host to host\|isolated" | tee -a $o/contexts.txt
done
