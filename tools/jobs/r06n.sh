#!/bin/bash
# round 6: default 4 slots with every slot's streams created with the context (main streams first): suite, first / second / third context, bench
o=gpurun_out/r06n; mkdir -p $o
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 | tee $o/gpu_tests.txt
echo "=== defaults (4 slots, 16 queues)" | tee -a $o/contexts_default.txt
timeout 700 python tools/second_context_stage_probe.py 2>&1 | grep "context\|own streams\|host to host\|isolated" | tee -a $o/contexts_default.txt
python bench.py --steps 20 --warmup 5 > $o/bench.json 2> $o/bench.err; python -c "
import json; d=json.loads(open('$o/bench.json').read().strip().splitlines()[-1]); print(d['value'], d['resident']['value'], d['end_to_end']['value'], d['single_proof_latency_ms'], d['config']['slots'], d['roofline_valu']['frac'], len(open('$o/bench.json').read()))"
