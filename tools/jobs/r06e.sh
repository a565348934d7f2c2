#!/bin/bash
# round 6, fifth GPU call: the suite after the removal of the NAF digits; the stream -> hardware-queue probe on the first / second / third
# context of a process (tools/second_context_stage_probe.py with masp_hip_ctx_stream_concurrency) at 4 slots / 24 queues and 3 / 16
o=gpurun_out/r06e; mkdir -p $o
timeout 2400 python -m pytest tests -m gpu -x -q > $o/gpu_tests.txt 2>&1; tail -3 $o/gpu_tests.txt
for cfg in "4 24" "3 16" "4 32"; do
  set -- $cfg
  echo "=== slots $1 queues $2" | tee -a $o/second_context_stream_concurrency.txt
  MASP_HIP_SLOTS=$1 GPU_MAX_HW_QUEUES=$2 timeout 600 python tools/second_context_stage_probe.py 2>&1 | grep -v Warning | tee -a $o/second_context_stream_concurrency.txt
done
