#!/bin/bash
# lone latency: var_mul on two quads (both builds); lanes of the narrow-window B2 accumulation 2^16 (default) vs 2^15
o=gpurun_out/r04w; mkdir -p $o
timeout 900 python -m pytest tests/test_gpu_endomorphism.py tests/test_golden_proofs.py -x -q 2>&1 | tail -3 | tee $o/tests.txt
for i in 1 2 3; do
  for L in masp_amd/libmasp_hip.so masp_amd/libmasp_hip_L15.so; do
    MASP_HIP_LIBRARY=$PWD/$L MASP_BENCH_E2E=0 python bench.py --steps 4 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$L', round(d['value'],1), round(d['resident']['value'],1), round(d['single_proof_latency']['host_to_host_ms'],3), round(d['single_proof_latency']['resident_witness_ms'],3))" | tee -a $o/ab.txt
  done
done
