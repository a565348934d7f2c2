#!/bin/bash
# quad tail kernels in a translation unit of their own (hip) vs without them at all (Q) vs the 10:35 build (B): same box, mixed order
o=gpurun_out/r04z3; mkdir -p $o
timeout 900 python -m pytest tests/test_golden_proofs.py tests/test_gpu_endomorphism.py -x -q 2>&1 | tail -2 | tee $o/tests.txt
for L in hip Q B hip B Q hip Q B hip; do
    lib=masp_amd/libmasp_hip.so; [ $L != hip ] && lib=masp_amd/libmasp_hip_$L.so
    v=$(MASP_HIP_LIBRARY=$PWD/$lib MASP_BENCH_E2E=0 python bench.py --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.1f  resident %.1f  gpu_ms %.2f lat %.2f' % (d['value'], d['resident']['value'], d['resident']['gpu_event_ms_per_step'], d['single_proof_latency_ms']))")
    echo "$L: $v" | tee -a $o/ab.txt
done
