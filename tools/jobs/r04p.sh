#!/bin/bash
# lone-proof launch graph: its tests, then host-to-host latency with the graph on / off on the same box
o=gpurun_out/r04p; mkdir -p $o
timeout 900 python -m pytest tests/test_gpu_lone_graph.py tests/test_gpu_parity.py::test_empty_and_ragged_job_lists -x -q 2>&1 | tail -15 > $o/tests.txt; cat $o/tests.txt
for i in 1 2; do
  for g in 1 -1; do
    MASP_HIP_LONE_GRAPH=$g MASP_BENCH_E2E=0 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('graph $g:', d['value'], json.dumps(d['single_proof_latency']['host_to_host_ms']), d['single_proof_latency']['resident_witness_one_by_one_launches_ms'], d['single_proof_latency']['graph_replays'])" | tee -a $o/ab.txt
  done
done
