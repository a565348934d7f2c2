#!/bin/bash
# round 6, tenth GPU call: lone-proof latency with the side streams created lazily (the build) against eagerly (as before), same box, alternating;
# the tests the last call missed
o=gpurun_out/r06j; mkdir -p $o
timeout 900 python -m pytest tests/test_golden_proofs.py tests/test_gpu_lone_and_warm.py tests/test_gpu_hw_queues.py tests/test_gpu_lone_graph.py tests/test_capi_harness.py -m gpu -x -q > $o/tests.txt 2>&1; tail -3 $o/tests.txt
for rep in 1 2 3; do
  for L in masp_amd/libmasp_hip.so tools/_build/ab/libmasp_hip_eager_aux.so; do
    echo "== $L" >> $o/lone_lazy_vs_eager_aux.txt
    MASP_HIP_LIBRARY=$PWD/$L LONE_CHAINS=1 LONE_ONLY=spend python tools/lone_sweep.py >> $o/lone_lazy_vs_eager_aux.txt 2>&1
  done
done
grep "lone ms\|==" $o/lone_lazy_vs_eager_aux.txt
for rep in 1 2; do
  for L in masp_amd/libmasp_hip.so tools/_build/ab/libmasp_hip_eager_aux.so; do
    v=$(MASP_HIP_LIBRARY=$PWD/$L MASP_BENCH_E2E=0 MASP_BENCH_OTHER=0 python bench.py --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('value %.1f resident %.1f lone %.2f / resident witness %.2f' % (d['value'], d['resident']['value'], d['single_proof_latency_ms'], d['single_proof_latency']['resident_witness_ms']))")
    echo "$L: $v" | tee -a $o/bench_lazy_vs_eager_aux.txt
  done
done
