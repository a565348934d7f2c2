#!/bin/bash
# round 6, sixth GPU call: what makes a LATER context of a process slower at 4 slots?  (a) device memory allocated, freed and allocated again
# (tools/realloc_gather_ubench); (b) the probe with less scratch at 4 slots and with more at 3; (c) a build whose contexts hand their
# released buffers to the next context instead of freeing them (-DMASP_KEEP_RELEASED_BUFFERS=1)
o=gpurun_out/r06f; mkdir -p $o
tools/_build/realloc_gather_ubench 176 > $o/realloc_gather_ubench.txt 2>&1; cat $o/realloc_gather_ubench.txt
run() { echo "=== $*" | tee -a $o/second_context_memory.txt; env "$@" timeout 600 python tools/second_context_stage_probe.py 2>&1 | grep "context\|own streams\|host to host" | tee -a $o/second_context_memory.txt; }
run MASP_HIP_SLOTS=4 GPU_MAX_HW_QUEUES=32 MASP_HIP_TREE_SUB=43
run MASP_HIP_SLOTS=3 GPU_MAX_HW_QUEUES=16 MASP_HIP_TREE_SUB=128
run MASP_HIP_SLOTS=4 GPU_MAX_HW_QUEUES=32 MASP_HIP_LIBRARY=$PWD/tools/_build/ab/libmasp_hip_keepbuf.so
