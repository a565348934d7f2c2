#!/bin/bash
# round 6: verifier streams owned by the context and kept off the batch streams' queues: contexts A / B / C (full probe output), bench twice, the verify + hw-queue tests
o=gpurun_out/r06s; mkdir -p $o
timeout 900 python -m pytest tests/test_gpu_verify.py tests/test_gpu_hw_queues.py tests/test_gpu_lone_and_warm.py tests/test_subgroup_checks.py -m gpu -x -q 2>&1 | tail -15 | tee $o/tests.txt
timeout 700 python tools/second_context_stage_probe.py 2>&1 | grep -v Warning | tee $o/contexts.txt
for rep in 1 2; do
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $o/bench$rep.json 2> $o/bench.err; python -c "
import json; d=json.loads(open('$o/bench$rep.json').read().strip().splitlines()[-1]); print('value %.1f resident %.1f e2e %.1f lone %.2f (resident witness %.2f) output lone %.2f convert lone %.2f slots %d valu %.3f' % (d['value'], d['resident']['value'], d['end_to_end']['value'], d['single_proof_latency_ms'], d['single_proof_latency']['resident_witness_ms'], d['other_circuits']['output']['single_proof_latency_ms'], d['other_circuits']['convert']['single_proof_latency_ms'], d['config']['slots'], d['roofline_valu']['frac']))" | tee -a $o/bench_summary.txt
done
