#!/bin/bash
o=gpurun_out/r04j; mkdir -p $o
timeout 1200 python -m pytest tests/test_golden_proofs.py tests/test_gpu_parity.py tests/test_gpu_batch_mode.py -m gpu -x -q 2>&1 | tail -4 > $o/tests.txt; cat $o/tests.txt
MASP_BENCH_E2E=0 python bench.py --steps 4 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('value', round(d['value'],1), 'lone latency ms', round(d['single_proof_latency_ms'],2))"
python tools/lone_timeline.py > $o/lone_timeline.txt 2>&1; tail -40 $o/lone_timeline.txt | cut -c1-150
