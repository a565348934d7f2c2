#!/bin/bash
o=gpurun_out/r04o; mkdir -p $o
tools/_build/fp28_mul_ubench | tee $o/fp28.txt
timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > $o/tests.txt; cat $o/tests.txt
MASP_BENCH_E2E=0 bash tools/ab.sh masp_amd/libmasp_hip_B.so masp_amd/libmasp_hip.so 3 > $o/ab.txt 2>&1; cat $o/ab.txt
