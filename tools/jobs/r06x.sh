#!/bin/bash
# round 6, the FINAL binary (one host entry point newer than r06z's): suite, bench at the driver's flags, the Spend-only kernel trace with its
# own bench line, PMC traffic (Spend batches only) and the VALU model — every tracked measurement stamped with this library's hash
o=gpurun_out/r06x; mkdir -p $o
sha256sum masp_amd/libmasp_hip.so | cut -c1-16 > $o/library_sha16.txt
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 > $o/gpu_tests.txt; cat $o/gpu_tests.txt
python bench.py --steps 20 --warmup 5 > $o/bench_driver_flags_steps20_warmup5.json 2> $o/bench.err; tail -c 300 $o/bench_driver_flags_steps20_warmup5.json
PROF_ARGS="--steps 4 --warmup 1 --no-cpu-baseline" bash tools/prof_run.sh r06x_spend_only MASP_HIP_SLOTS=1 MASP_BENCH_OTHER=0 MASP_BENCH_E2E=0 MASP_BENCH_LONE=0 > $o/prof_spend_only.log 2>&1
db=$(find gpurun_out/prof_r06x_spend_only -name "*.db" | head -1)
python tools/g1_stage_stats.py $db > $o/g1_stage_per_spend_msm.txt 2>&1
cp gpurun_out/prof_r06x_spend_only/batch.txt $o/kernel_stats_spend_batches_only_grid_y_256.txt
grep '^{' gpurun_out/prof_r06x_spend_only/bench.log | tail -1 > $o/g1_stage_per_spend_msm_bench_line_of_the_same_run.json
PMC_OUT=r06x/pmc_traffic bash tools/pmc_traffic.sh > $o/pmc_traffic.log 2>&1
bash tools/valu_model.sh gpurun_out/r06x/valu_model.json > $o/valu_model.log 2>&1; head -1 $o/valu_model.log | cut -c1-300
rm -rf gpurun_out/prof_r06x_spend_only gpurun_out/pmc gpurun_out/pmc_valu
ls $o
