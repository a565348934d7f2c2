#!/bin/bash
# round 6, the FINAL binary (= r06w's + no upload chain under lone_proof_graph, HIP error text in "H2D copy failed"): the lone-graph stress first,
# then everything tracked under profiles/ re-measured on it, in one call
o=gpurun_out/r06ze; mkdir -p $o
timeout 600 python tools/lone_graph_stress.py 40 > $o/lone_graph_stress.txt 2>&1; tail -3 $o/lone_graph_stress.txt
bash tools/round_evidence.sh r06ze > gpurun_out/r06ze_evidence.log 2>&1; tail -4 gpurun_out/r06ze_evidence.log | cut -c1-300
PMC_OUT=r06ze/pmc_traffic bash tools/pmc_traffic.sh > $o/pmc_traffic.log 2>&1
bash tools/valu_model.sh $o/valu_model.json > $o/valu_model.log 2>&1; head -1 $o/valu_model.log | cut -c1-300
python bench.py --in-library --gpus 1 --steps 8 --warmup 2 > $o/bench_in_library_1_gpu.json 2>> $o/bench.err
MASP_BENCH_CIRCUIT=mixed python bench.py --steps 8 --warmup 2 --no-cpu-baseline > $o/bench_mixed_workload.json 2>> $o/bench.err
python tools/cxx_tx_prover_bench.py 5120 > $o/cxx_tx_prover_5120.txt 2>&1; tail -3 $o/cxx_tx_prover_5120.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $o/smoke.txt 2>&1; tail -1 $o/smoke.txt
rm -rf gpurun_out/pmc gpurun_out/pmc_valu
ls $o
