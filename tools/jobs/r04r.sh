#!/bin/bash
# AMD_DIRECT_DISPATCH=0 (kernel launches handed to the runtime's command thread) against the default: same box, three rounds, end to end included
sel='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("%.1f  resident %.1f  end_to_end %.1f  lat %.2f" % (d["value"], d["resident"]["value"], d["end_to_end"]["value"], d["single_proof_latency_ms"]))'
for r in 1 2 3; do
  for cfg in "DEFAULT=1" "AMD_DIRECT_DISPATCH=0"; do
    echo "$cfg: $(env $cfg python bench.py --steps 12 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "$sel")"
  done
done
for w in output convert; do for cfg in "DEFAULT=1" "AMD_DIRECT_DISPATCH=0"; do
  echo "$w $cfg: $(env $cfg MASP_BENCH_CIRCUIT=$w python bench.py --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("%.1f  resident %.1f  lat %.2f" % (d["value"], d["resident"]["value"], d["single_proof_latency_ms"]))')"
done; done
