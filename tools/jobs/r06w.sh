#!/bin/bash
# round 6, the FINAL binary (4 slots, 15 streams per context, chained uploads): everything tracked under profiles/ re-measured on it, in one call
bash tools/round_evidence.sh r06w > gpurun_out/r06w_evidence.log 2>&1; tail -4 gpurun_out/r06w_evidence.log | cut -c1-300
PMC_OUT=r06w/pmc_traffic bash tools/pmc_traffic.sh > gpurun_out/r06w/pmc_traffic.log 2>&1
bash tools/valu_model.sh gpurun_out/r06w/valu_model.json > gpurun_out/r06w/valu_model.log 2>&1; head -1 gpurun_out/r06w/valu_model.log | cut -c1-300
python bench.py --in-library --gpus 1 --steps 8 --warmup 2 > gpurun_out/r06w/bench_in_library_1_gpu.json 2>> gpurun_out/r06w/bench.err
MASP_BENCH_CIRCUIT=mixed python bench.py --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/r06w/bench_mixed_workload.json 2>> gpurun_out/r06w/bench.err
rm -rf gpurun_out/pmc gpurun_out/pmc_valu
ls gpurun_out/r06w
