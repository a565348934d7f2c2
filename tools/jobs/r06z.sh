#!/bin/bash
# round 6, the final build's evidence in ONE call (same box for all of it): tools/round_evidence.sh (GPU suite, bench at the driver's flags, kernel
# statistics incl. the Spend-only per-MSM table, lone timeline, SQ counters, micro-benchmarks), then the PMC traffic passes and the VALU model
bash tools/round_evidence.sh r06z > gpurun_out/r06z_evidence.log 2>&1; tail -5 gpurun_out/r06z_evidence.log
PMC_OUT=r06z/pmc_traffic bash tools/pmc_traffic.sh > gpurun_out/r06z/pmc_traffic.log 2>&1; tail -3 gpurun_out/r06z/pmc_traffic.log | cut -c1-300
bash tools/valu_model.sh gpurun_out/r06z/valu_model.json > gpurun_out/r06z/valu_model.log 2>&1; head -2 gpurun_out/r06z/valu_model.log | cut -c1-400
rm -rf gpurun_out/pmc gpurun_out/pmc_valu
ls gpurun_out/r06z
