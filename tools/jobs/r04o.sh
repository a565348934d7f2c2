#!/bin/bash
# runtime knobs of the HIP / ROCr stack against the defaults (16 hardware queues): same box, two rounds
sel='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("%.1f  resident %.1f  end_to_end %.1f  lat %.2f" % (d["value"], d["resident"]["value"], d["end_to_end"]["value"], d["single_proof_latency_ms"]))'
for r in 1 2; do
  for cfg in "DEFAULT=1" "HIP_FORCE_DEV_KERNARG=1" "HIP_FORCE_DEV_KERNARG=0" "HSA_ENABLE_SDMA=0" "GPU_MAX_HW_QUEUES=32" "HIP_LAUNCH_BLOCKING=0 AMD_DIRECT_DISPATCH=1" "AMD_DIRECT_DISPATCH=0"; do
    echo "$cfg: $(env $cfg python bench.py --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "$sel")"
  done
done
