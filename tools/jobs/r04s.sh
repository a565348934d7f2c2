#!/bin/bash
# bench.py's end_to_end region with and without the collection + freeze of the Python heap before it, 3 and 4 slots: same box
sel='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("%.1f  resident %.1f  end_to_end %.1f" % (d["value"], d["resident"]["value"], d["end_to_end"]["value"]))'
for r in 1 2; do
  for cfg in "MASP_BENCH_GC_FREEZE=0 MASP_HIP_SLOTS=3 GPU_MAX_HW_QUEUES=16" "MASP_BENCH_GC_FREEZE=1 MASP_HIP_SLOTS=3 GPU_MAX_HW_QUEUES=16" "MASP_BENCH_GC_FREEZE=0 MASP_HIP_SLOTS=4 GPU_MAX_HW_QUEUES=24" "MASP_BENCH_GC_FREEZE=1 MASP_HIP_SLOTS=4 GPU_MAX_HW_QUEUES=24"; do
    echo "$cfg: $(env $cfg python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "$sel")"
  done
done
