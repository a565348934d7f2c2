#!/bin/bash
# round 6: host-to-device copies of concurrent calls chained one batch after the other (the build) against as they come (variant), same box
o=gpurun_out/r06k; mkdir -p $o
timeout 600 python -m pytest tests/test_gpu_batch_mode.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -2 | tee $o/tests.txt
for rep in 1 2 3 4; do
  for L in masp_amd/libmasp_hip.so tools/_build/ab/libmasp_hip_nochain.so; do
    v=$(MASP_HIP_LIBRARY=$PWD/$L MASP_BENCH_E2E=0 MASP_BENCH_OTHER=0 MASP_BENCH_LONE=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('value %.1f resident %.1f ratio %.4f' % (d['value'], d['resident']['value'], d['value']/d['resident']['value']))")
    echo "$L: $v" | tee -a $o/upload_chain_ab.txt
  done
done
for rep in 1 2; do
  for L in masp_amd/libmasp_hip.so tools/_build/ab/libmasp_hip_nochain.so; do
    v=$(MASP_HIP_LIBRARY=$PWD/$L MASP_BENCH_OTHER=0 MASP_BENCH_LONE=0 python bench.py --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('8 steps: value %.1f resident %.1f e2e %.1f' % (d['value'], d['resident']['value'], d['end_to_end']['value']))")
    echo "$L: $v" | tee -a $o/upload_chain_ab.txt
  done
done
