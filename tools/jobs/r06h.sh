#!/bin/bash
# round 6, eighth GPU call: the first context of a process whose runtime already holds its full pool of hardware queues (64 streams created,
# used and destroyed first), against the size of that pool: GPU_MAX_HW_QUEUES 8 ... 32, 3 slots; and the same without the stream history
o=gpurun_out/r06h; mkdir -p $o
run() { echo "=== $*" | tee -a $o/queue_pool_size.txt; env "$@" timeout 500 python tools/second_context_stage_probe.py 2>&1 | grep "context\|own streams\|host to host\|isolated\|streams created" | tee -a $o/queue_pool_size.txt; }
for q in 8 12 16 20 24 32; do run GPU_MAX_HW_QUEUES=$q SCP_PRE_STREAMS=64 SCP_CONTEXTS=1 MASP_HIP_SLOTS=3; done
for q in 16 32; do run GPU_MAX_HW_QUEUES=$q SCP_CONTEXTS=1 MASP_HIP_SLOTS=3; done
