#!/bin/bash
# A/B sweep of the runtime knobs inside ONE gpurun call (box-to-box variance is ~25 %).
# usage: tools/sweep.sh "VAR=val VAR2=val" "..." ...
for cfg in "$@"; do
  out=$(env $cfg timeout 300 python bench.py --steps ${SWEEP_STEPS:-32} --warmup 4 --no-cpu-baseline 2>&1 | tail -1)
  echo "$cfg => $(echo "$out" | python -c 'import sys,json
try:
    d=json.loads(sys.stdin.read()); print("%.1f proofs/s  %.3f ms/proof  lat %.1f ms  acc %.2f ms" % (d["value"], d["ms_per_proof"], d["single_proof_latency_ms"], d["roofline"].get("avg_launch_ms", -1)))
except Exception as e: print("ERR", e)')"
done
