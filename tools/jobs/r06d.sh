#!/bin/bash
# round 6, fourth GPU call: does the slower second context of round 4 (profiles/r04e_second_context_in_a_process.txt) reproduce — with round 4's
# library (commit 692a699, rebuilt) and with today's, at round 4's 4 slots / 24 queues and at the default 3 / 16?
o=gpurun_out/r06d; mkdir -p $o
for L in tools/_build/ab/libmasp_hip_r04e_692a699.so masp_amd/libmasp_hip.so; do
  for cfg in "4 24" "3 16"; do
    set -- $cfg
    echo "=== $L slots $1 queues $2" | tee -a $o/second_context_old_and_new.txt
    MASP_HIP_LIBRARY=$PWD/$L MASP_HIP_SLOTS=$1 GPU_MAX_HW_QUEUES=$2 timeout 600 python tools/second_context_stage_probe.py 2>&1 | grep -v Warning | tee -a $o/second_context_old_and_new.txt
  done
done
