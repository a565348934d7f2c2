#!/bin/bash
# The measurements a round's figures come from, in ONE gpurun call (same box for all of them): the GPU parity suite, the bench
# line at 20 steps, rocprofv3 kernel statistics with one batch in flight and with the default three, the lone-proof timeline,
# the SQ counters per kernel and the product / instruction-rate micro-benchmarks.  usage: tools/round_evidence.sh <tag>
tag=${1:-evidence}
o=gpurun_out/$tag
mkdir -p $o
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 > $o/gpu_tests.txt
python bench.py --steps 20 --warmup 5 > $o/bench_driver_flags_steps20_warmup5.json 2> $o/bench.err
PROF_ARGS="--steps 2 --warmup 1 --no-cpu-baseline" bash tools/prof_run.sh ${tag}_slots1 MASP_HIP_SLOTS=1 > $o/prof_slots1.log 2>&1
cp gpurun_out/prof_${tag}_slots1/all.txt $o/kernel_stats_one_slot_serialized_all_dispatches.txt
cp gpurun_out/prof_${tag}_slots1/batch.txt $o/kernel_stats_one_slot_serialized_batches_of_256_only.txt
db=$(find gpurun_out/prof_${tag}_slots1 -name "*.db" | head -1)
python tools/lone_timeline.py $db > $o/lone_proof_timeline.txt 2>&1
# the G1 bucket stage per Spend MSM: a run of full Spend batches only, one in flight (tools/g1_stage_stats.py), and the bench line of the
# SAME run next to it (its roofline.avg_launch_ms / kernel_ms_per_launch are what the table is set against)
PROF_ARGS="--steps 4 --warmup 1 --no-cpu-baseline" bash tools/prof_run.sh ${tag}_spend_only MASP_HIP_SLOTS=1 MASP_BENCH_OTHER=0 MASP_BENCH_E2E=0 MASP_BENCH_LONE=0 > $o/prof_spend_only.log 2>&1
db=$(find gpurun_out/prof_${tag}_spend_only -name "*.db" | head -1)
python tools/g1_stage_stats.py $db > $o/g1_stage_per_spend_msm.txt 2>&1
cp gpurun_out/prof_${tag}_spend_only/batch.txt $o/kernel_stats_spend_batches_only_grid_y_256.txt
grep '^{' gpurun_out/prof_${tag}_spend_only/bench.log | tail -1 > $o/g1_stage_per_spend_msm_bench_line_of_the_same_run.json
PROF_ARGS="--steps 3 --warmup 1 --no-cpu-baseline" bash tools/prof_run.sh ${tag}_default > $o/prof_default.log 2>&1
cp gpurun_out/prof_${tag}_default/all.txt $o/kernel_stats_default_bench_all_dispatches.txt
rm -rf gpurun_out/prof_${tag}_slots1 gpurun_out/prof_${tag}_default gpurun_out/prof_${tag}_spend_only
PMC_OUT=$tag/pmc_sq_kernels_one_slot bash tools/pmc_sq_kernels.sh > $o/pmc_sq.log 2>&1
rm -rf gpurun_out/pmc_sqk
bash tools/build_tools.sh > /dev/null 2>&1
tools/_build/ubench > $o/instruction_rates_and_products_ubench.txt 2>&1
tools/_build/valu_rate_ubench > $o/valu_instruction_cost_classes_ubench.txt 2>&1
sha256sum masp_amd/libmasp_hip.so | cut -c1-16 > $o/library_sha16.txt
tools/_build/ntt_ubench > $o/ntt_ubench.txt 2>&1
cat $o/gpu_tests.txt; tail -c 300 $o/bench_driver_flags_steps20_warmup5.json; ls -la $o
