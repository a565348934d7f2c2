#!/bin/bash
# A/B of two option settings of ONE build on the SAME box: usage: tools/ab_env.sh "<env A>" "<env B>" [reps=3] [bench args...]
# e.g. tools/ab_env.sh "MASP_HIP_DIGITS=-1" "MASP_HIP_DIGITS=0" 3   — alternates A, B, A, B ... and prints proofs/s of each run
A=$1; B=$2; reps=${3:-3}; shift 3
for i in $(seq $reps); do
  for L in "$A" "$B"; do
    v=$(env $L python bench.py --steps 8 --warmup 2 --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline'].get('kernel_ms_per_launch', {}); o=d.get('other_circuits', {}); print('%.1f  resident %.1f  e2e %.1f  lat %.2f  stage %.2f ms (pass1 %.2f pass2 %.2f pts %.2f)  output %.0f convert %.0f  sclk %s W %s' % (d['value'], d['resident']['value'], d['end_to_end']['value'], d['single_proof_latency_ms'], d['roofline']['avg_launch_ms'], k.get('k_tree_pass1', 0), k.get('k_tree_pass2', 0), k.get('k_msm_accumulate_pts', 0), o.get('output', {}).get('value', 0), o.get('convert', {}).get('value', 0), d.get('roofline_valu', {}).get('sclk_mhz'), d.get('roofline_valu', {}).get('socket_power_w')))")
    echo "$L: $v"
  done
done
