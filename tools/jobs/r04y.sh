#!/bin/bash
# is the ~2 % between the 10:35 build (B) and now real?  mixed order, 8 steps each
o=gpurun_out/r04y; mkdir -p $o
for L in hip B hip B B hip hip B B hip; do
    v=$(MASP_HIP_LIBRARY=$PWD/masp_amd/$( [ $L = hip ] && echo libmasp_hip.so || echo libmasp_hip_B.so ) MASP_BENCH_E2E=0 python bench.py --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.1f  resident %.1f  gpu_ms %.2f lat %.2f' % (d['value'], d['resident']['value'], d['resident']['gpu_event_ms_per_step'], d['single_proof_latency_ms']))")
    echo "$L: $v" | tee -a $o/ab.txt
done
