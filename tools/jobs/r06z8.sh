#!/bin/bash
# the C++ mirror's batch path: synthesis threads 16 (all the CPUs) / 12 / 8 next to the proving threads; per-batch trace at 12
o=gpurun_out/r06z8; mkdir -p $o
for t in 0 12 8; do MASP_TXP_THREADS=$t python tools/cxx_tx_prover_bench.py 5120 > $o/threads_$t.txt 2>&1; echo "threads $t: $(grep timed $o/threads_$t.txt)"; done
MASP_TXP_THREADS=12 MASP_TXP_TRACE=1 python tools/cxx_tx_prover_bench.py 2048 > $o/trace_12.txt 2>&1; tail -9 $o/trace_12.txt
