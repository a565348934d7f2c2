#!/bin/bash
# round 6: the order in which a context creates its slots' streams — slot 0's five, other mains, other side streams (the build) / all mains first / slot by slot:
# lone-proof latency inside bench.py and batch throughput, same box, alternating; then first / second / third context with the build
o=gpurun_out/r06o; mkdir -p $o
for rep in 1 2 3; do
  for L in masp_amd/libmasp_hip.so tools/_build/ab/libmasp_hip_order1.so tools/_build/ab/libmasp_hip_order2.so; do
    v=$(MASP_HIP_LIBRARY=$PWD/$L MASP_BENCH_E2E=0 MASP_BENCH_OTHER=0 python bench.py --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('value %.1f resident %.1f lone %.2f / resident witness %.2f' % (d['value'], d['resident']['value'], d['single_proof_latency_ms'], d['single_proof_latency']['resident_witness_ms']))")
    echo "$L: $v" | tee -a $o/stream_order_ab.txt
  done
done
echo "=== the build, defaults" | tee -a $o/contexts.txt
timeout 700 python tools/second_context_stage_probe.py 2>&1 | grep "context\|own streams\|host to host\|isolated" | tee -a $o/contexts.txt
