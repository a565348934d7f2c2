#!/bin/bash
# same-box check that today's changes did not touch batch throughput: build of 10:35 (before the endomorphism / quads / allocation counters) vs now
o=gpurun_out/r04x; mkdir -p $o
MASP_BENCH_E2E=0 bash tools/ab.sh masp_amd/libmasp_hip_B.so masp_amd/libmasp_hip.so 3 > $o/ab.txt 2>&1; cat $o/ab.txt
