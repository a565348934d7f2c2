#!/bin/bash
# r04b: hand-written carry chains for add / sub / neg / dbl / reduce (field.cuh) + sliced planes + version scripts + MASP_LAUNCH
# against the round-3 kernels (libmasp_hip_A.so); the 13 x 30-bit product ubench; SQ counters of the big kernels
o=gpurun_out/r04b; mkdir -p $o
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > $o/tests.txt
cat $o/tests.txt
MASP_BENCH_E2E=0 bash tools/ab.sh masp_amd/libmasp_hip_A.so masp_amd/libmasp_hip.so 2 > $o/ab.txt 2>&1
cat $o/ab.txt
tools/_build/fp30_mul_ubench > $o/fp30.txt 2>&1; cat $o/fp30.txt
tools/_build/fp28_mul_ubench > $o/fp28.txt 2>&1; cat $o/fp28.txt
PROF_ARGS="--steps 2 --warmup 1 --no-cpu-baseline" PROF_GY=256 MASP_BENCH_E2E=0 bash tools/prof_run.sh r04b_slots1 MASP_HIP_SLOTS=1 > $o/prof_slots1.txt 2>&1
head -45 gpurun_out/prof_r04b_slots1/all.txt | cut -c1-140
PMC_OUT=r04b_pmc_sq bash tools/pmc_sq_kernels.sh > $o/pmc_sq.txt 2>&1; tail -30 $o/pmc_sq.txt
