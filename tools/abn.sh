#!/bin/bash
# A/B/C... of several builds of libmasp_hip on the SAME box: usage: tools/abn.sh <reps> <lib.so> <lib.so> ...   (paths from the repository root: the build under test is masp_amd/libmasp_hip.so,
# earlier builds are kept under tools/_build/ab/ — masp_amd/libmasp_hip_*.so does not travel to the GPU box, .gpurunignore)
# prints proofs/s and the isolated G1 bucket stage by kernel group (repeats within +-0.3 % on one box) of every run
reps=$1; shift
for r in $(seq $reps); do
  for L in "$@"; do
    v=$(MASP_BENCH_E2E=0 MASP_BENCH_OTHER=0 MASP_HIP_LIBRARY=$PWD/$L python bench.py --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['kernel_ms_per_launch']; print('%.1f  resident %.1f  lat %.2f  stage %.2f  pass1 %.2f pass2 %.2f pts %.2f' % (d['value'], d['resident']['value'], d['single_proof_latency_ms'], d['roofline']['avg_launch_ms'], k['k_tree_pass1'], k['k_tree_pass2'], k['k_msm_accumulate_pts']))")
    echo "$L: $v"
  done
done
