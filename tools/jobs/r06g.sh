#!/bin/bash
# round 6, seventh GPU call: later contexts at 4 slots are 5 % slower — is it the context's own slot count or the process's history
# (SCP_SLOTS sequences), is it the number of streams the process has created (SCP_PRE_STREAMS), are the kernels themselves slower (isolated stage)?
o=gpurun_out/r06g; mkdir -p $o
run() { echo "=== $*" | tee -a $o/second_context_history.txt; env "$@" timeout 700 python tools/second_context_stage_probe.py 2>&1 | grep "context\|own streams\|host to host\|isolated\|streams created" | tee -a $o/second_context_history.txt; }
run GPU_MAX_HW_QUEUES=32 SCP_SLOTS=4,3,4
run GPU_MAX_HW_QUEUES=32 SCP_SLOTS=3,4,4
run GPU_MAX_HW_QUEUES=32 SCP_SLOTS=4,4,4 SCP_PRE_STREAMS=48
