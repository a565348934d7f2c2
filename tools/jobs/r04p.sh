#!/bin/bash
# slots x hardware queues again, now that the slots' streams no longer share queues (3 slots / 16 queues is the default): same box, two rounds
sel='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("%.1f  resident %.1f" % (d["value"], d["resident"]["value"]))'
for r in 1 2; do
  for cfg in "MASP_HIP_SLOTS=3 GPU_MAX_HW_QUEUES=16" "MASP_HIP_SLOTS=4 GPU_MAX_HW_QUEUES=24 MASP_HIP_TREE_SUB=64" "MASP_HIP_SLOTS=4 GPU_MAX_HW_QUEUES=24 MASP_HIP_TREE_SUB=86" "MASP_HIP_SLOTS=5 GPU_MAX_HW_QUEUES=32 MASP_HIP_TREE_SUB=64" "MASP_HIP_SLOTS=2 GPU_MAX_HW_QUEUES=16 MASP_HIP_TREE_SUB=128" "MASP_HIP_SLOTS=3 GPU_MAX_HW_QUEUES=16 MASP_HIP_TREE_SUB=128"; do
    echo "$cfg: $(env $cfg MASP_BENCH_E2E=0 python bench.py --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "$sel")"
  done
done
