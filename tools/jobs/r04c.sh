#!/bin/bash
# r04c: slope numerators pre-multiplied in pass 1 (MASP_TREE_QNUM=1, libmasp_hip.so) against the build before it (libmasp_hip_B.so)
o=gpurun_out/r04c; mkdir -p $o
timeout 1200 python -m pytest tests/test_gpu_bucket_tree.py tests/test_golden_proofs.py tests/test_gpu_batch_mode.py tests/test_subgroup_checks.py tests/test_gpu_configs4.py -m gpu -x -q 2>&1 | tail -5 > $o/tests.txt
cat $o/tests.txt
MASP_BENCH_E2E=0 bash tools/ab.sh masp_amd/libmasp_hip_B.so masp_amd/libmasp_hip.so 2 > $o/ab.txt 2>&1
cat $o/ab.txt
PROF_ARGS="--steps 2 --warmup 1 --no-cpu-baseline" PROF_GY=256 MASP_BENCH_E2E=0 bash tools/prof_run.sh r04c_slots1 MASP_HIP_SLOTS=1 > $o/prof_slots1.txt 2>&1
head -24 gpurun_out/prof_r04c_slots1/all.txt | cut -c1-140
