#!/bin/bash
# 3 slots / 16 queues (default) against 4 slots / 24 queues at the driver's own flags (--steps 20 --warmup 5), end to end included: same box, two rounds
sel='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("%.1f  resident %.1f  end_to_end %.1f  lat %.2f" % (d["value"], d["resident"]["value"], d["end_to_end"]["value"], d["single_proof_latency_ms"]))'
for r in 1 2; do
  for cfg in "MASP_HIP_SLOTS=3 GPU_MAX_HW_QUEUES=16" "MASP_HIP_SLOTS=4 GPU_MAX_HW_QUEUES=24" "MASP_HIP_SLOTS=4 GPU_MAX_HW_QUEUES=24 MASP_BENCH_H2H_CALLS=5"; do
    echo "$cfg: $(env $cfg python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "$sel")"
  done
done
