#!/bin/bash
# lone-proof timeline after the endomorphism and the quad tails
o=gpurun_out/r04u; mkdir -p $o
PROF_ARGS="--steps 2 --warmup 1 --no-cpu-baseline" PROF_GY=256 MASP_BENCH_E2E=0 bash tools/prof_run.sh r04u > $o/prof.txt 2>&1
python tools/lone_timeline.py $(find gpurun_out/prof_r04u -name "*.db" | head -1) > $o/lone_timeline.txt 2>&1
tail -3 $o/lone_timeline.txt
rm -f gpurun_out/prof_r04u/*.db
