#!/bin/bash
# tree sub-batch 86 (default) vs 128 vs 256-with-2-slots on the same box, 8 steps, mixed order
o=gpurun_out/r04z4; mkdir -p $o
for V in "MASP_HIP_TREE_SUB=86" "MASP_HIP_TREE_SUB=128" "MASP_HIP_TREE_SUB=86" "MASP_HIP_TREE_SUB=128" "MASP_HIP_TREE_SUB=128" "MASP_HIP_TREE_SUB=86"; do
    v=$(env $V MASP_BENCH_E2E=0 python bench.py --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.1f  resident %.1f  gpu_ms %.2f stage_ms %.2f frac %.5f' % (d['value'], d['resident']['value'], d['resident']['gpu_event_ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac']))")
    echo "$V: $v" | tee -a $o/ab.txt
done
rocm-smi --showmeminfo vram 2>/dev/null | tail -3
