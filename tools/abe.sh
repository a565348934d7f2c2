#!/bin/bash
# A/B of builds of libmasp_hip on the SAME box INCLUDING the end-to-end region: usage: tools/abe.sh <reps> <lib.so> ...  (paths below masp_amd/)
reps=$1; shift
for r in $(seq $reps); do
  for L in "$@"; do
    v=$(MASP_HIP_LIBRARY=$PWD/masp_amd/$L python bench.py --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.1f  resident %.1f  end_to_end %.1f  lat %.2f' % (d['value'], d['resident']['value'], d['end_to_end']['value'], d['single_proof_latency_ms']))")
    echo "$L: $v"
  done
done
