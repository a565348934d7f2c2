#!/bin/bash
# round 6, after the evidence run: PMC traffic over Spend batches only, the Spend-only kernel trace with the bench line of the SAME run, the bench at
# the driver's flags with the clock watch on this process's own card, and the in-library line
o=gpurun_out/r06y; mkdir -p $o
PMC_OUT=r06y/pmc_traffic bash tools/pmc_traffic.sh > $o/pmc_traffic.log 2>&1; tail -3 $o/pmc_traffic.log | cut -c1-200
PROF_ARGS="--steps 4 --warmup 1 --no-cpu-baseline" bash tools/prof_run.sh r06y_spend_only MASP_HIP_SLOTS=1 MASP_BENCH_OTHER=0 MASP_BENCH_E2E=0 MASP_BENCH_LONE=0 > $o/prof_spend_only.log 2>&1
db=$(find gpurun_out/prof_r06y_spend_only -name "*.db" | head -1)
python tools/g1_stage_stats.py $db > $o/g1_stage_per_spend_msm.txt 2>&1
grep '^{' gpurun_out/prof_r06y_spend_only/bench.log | tail -1 > $o/g1_stage_per_spend_msm_bench_line_of_the_same_run.json
rm -rf gpurun_out/prof_r06y_spend_only gpurun_out/pmc
python bench.py --steps 20 --warmup 5 > $o/bench_driver_flags_steps20_warmup5.json 2> $o/bench.err; tail -c 600 $o/bench_driver_flags_steps20_warmup5.json
python bench.py --in-library --gpus 1 --steps 8 --warmup 2 > $o/bench_in_library_1_gpu.json 2>> $o/bench.err; cut -c1-200 $o/bench_in_library_1_gpu.json
MASP_BENCH_CIRCUIT=mixed python bench.py --steps 8 --warmup 2 --no-cpu-baseline > $o/bench_mixed_workload.json 2>> $o/bench.err; cut -c1-160 $o/bench_mixed_workload.json
