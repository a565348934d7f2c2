#!/bin/bash
# A/B of two builds of libmasp_hip on the SAME box (box-to-box variance is larger than most kernel changes):
# usage: tools/ab.sh <libA.so> <libB.so> [reps=3] [bench args...]   — alternates A, B, A, B ... and prints proofs/s of each run
A=$1; B=$2; reps=${3:-3}; shift 3
for i in $(seq $reps); do
  for L in $A $B; do
    v=$(MASP_HIP_LIBRARY=$PWD/$L python bench.py --steps 8 --warmup 2 --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline'].get('kernel_ms_per_launch', {}); print('%.1f  resident %.1f  lat %.2f  stage %.2f ms (pass1 %.2f pass2 %.2f pts %.2f)' % (d['value'], d['resident']['value'], d['single_proof_latency_ms'], d['roofline']['avg_launch_ms'], k.get('k_tree_pass1', 0), k.get('k_tree_pass2', 0), k.get('k_msm_accumulate_pts', 0)))")
    echo "$L: $v"
  done
done
