#!/bin/bash
# round 6: main streams measured and separated at context creation (slot-by-slot stream order): suite, contexts A / B / C, bench at the driver's flags twice
o=gpurun_out/r06r; mkdir -p $o
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 | tee $o/gpu_tests.txt
echo "=== the build, defaults" | tee -a $o/contexts.txt
timeout 700 python tools/second_context_stage_probe.py 2>&1 | grep "context\|own streams\|host to host\|isolated" | tee -a $o/contexts.txt
for rep in 1 2; do
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $o/bench$rep.json 2> $o/bench.err; python -c "
import json; d=json.loads(open('$o/bench$rep.json').read().strip().splitlines()[-1]); print('value %.1f resident %.1f e2e %.1f lone %.2f (resident witness %.2f) output lone %.2f convert lone %.2f slots %d valu %.3f' % (d['value'], d['resident']['value'], d['end_to_end']['value'], d['single_proof_latency_ms'], d['single_proof_latency']['resident_witness_ms'], d['other_circuits']['output']['single_proof_latency_ms'], d['other_circuits']['convert']['single_proof_latency_ms'], d['config']['slots'], d['roofline_valu']['frac']))" | tee -a $o/bench_summary.txt
done
