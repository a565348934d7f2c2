#!/bin/bash
# r04a: tree planes cut into 16-byte slices (MASP_TREE_SLICED=1, libmasp_hip.so) against whole elements (libmasp_hip_A.so)
o=gpurun_out/r04a; mkdir -p $o
timeout 900 python -m pytest tests/test_gpu_bucket_tree.py tests/test_golden_proofs.py tests/test_gpu_batch_mode.py -m gpu -x -q 2>&1 | tail -5 > $o/tests.txt
cat $o/tests.txt
MASP_BENCH_E2E=0 bash tools/ab.sh masp_amd/libmasp_hip_A.so masp_amd/libmasp_hip.so 2 > $o/ab.txt 2>&1
cat $o/ab.txt
PMC_STEPS=2 PMC_OUT=r04a_pmc_sliced MASP_BENCH_E2E=0 bash tools/pmc_traffic.sh > $o/pmc.log 2>&1
PROF_ARGS="--steps 2 --warmup 1 --no-cpu-baseline" PROF_GY=256 MASP_BENCH_E2E=0 bash tools/prof_run.sh r04a_slots1 MASP_HIP_SLOTS=1 > $o/prof_slots1.txt 2>&1
head -30 $o/prof_slots1.txt
