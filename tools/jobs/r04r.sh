#!/bin/bash
# endomorphism in k_groth16_var_mul: parity (both paths), the whole GPU suite, lone latency / value on the same box vs the build before (libmasp_hip_B.so)
o=gpurun_out/r04r; mkdir -p $o
timeout 900 python -m pytest tests/test_gpu_endomorphism.py -x -q 2>&1 | tail -15 > $o/endo.txt; cat $o/endo.txt
timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > $o/tests.txt; cat $o/tests.txt
for i in 1 2; do
  for L in masp_amd/libmasp_hip_B.so masp_amd/libmasp_hip.so; do
    MASP_HIP_LIBRARY=$PWD/$L MASP_BENCH_E2E=0 python bench.py --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$L', round(d['value'],1), round(d['resident']['value'],1), d['single_proof_latency_ms'], d.get('single_proof_latency'))" | tee -a $o/ab.txt
  done
done
