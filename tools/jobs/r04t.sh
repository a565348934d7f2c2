#!/bin/bash
# G1 tails of a lone proof over quads: whole GPU suite, then lone latency / value
o=gpurun_out/r04t; mkdir -p $o
timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > $o/tests.txt; cat $o/tests.txt
for i in 1 2 3; do
    MASP_BENCH_E2E=0 python bench.py --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), round(d['resident']['value'],1), d['single_proof_latency']['host_to_host_ms'], d['single_proof_latency']['resident_witness_ms'])" | tee -a $o/ab.txt
done
