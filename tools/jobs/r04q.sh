#!/bin/bash
# how long does the host take to enqueue a lone proof?  (timing build of prover.hip: -DMASP_ENQ_TIMING)
o=gpurun_out/r04q; mkdir -p $o
MASP_HIP_LIBRARY=$PWD/masp_amd/libmasp_hip_T.so MASP_BENCH_E2E=0 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2> $o/err.txt | tail -1 > $o/bench.json
grep "\[enq\] 1 proofs" $o/err.txt | tail -14 | tee $o/enq.txt
python -c "
import json; d=json.load(open('$o/bench.json')); print(d['value'], d['single_proof_latency'])" | tee -a $o/enq.txt
timeout 600 python -m pytest tests/test_gpu_lone_graph.py -x -q 2>&1 | tail -3 | tee -a $o/enq.txt
