#!/bin/bash
# which of today's changes, if any, costs the "high mode" of a bench run?  B = 10:35 build, E = now without the endomorphism in batch mode,
# Q = now without the quad tail kernels (smaller code object), hip = now
o=gpurun_out/r04z2; mkdir -p $o
for L in hip E Q B E B hip Q B Q E hip; do
    lib=masp_amd/libmasp_hip.so; [ $L != hip ] && lib=masp_amd/libmasp_hip_$L.so
    v=$(MASP_HIP_LIBRARY=$PWD/$lib MASP_BENCH_E2E=0 python bench.py --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.1f  resident %.1f  gpu_ms %.2f lat %.2f' % (d['value'], d['resident']['value'], d['resident']['gpu_event_ms_per_step'], d['single_proof_latency_ms']))")
    echo "$L: $v" | tee -a $o/ab.txt
done
