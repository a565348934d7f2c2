#!/bin/bash
run() { v=$(env "$@" python bench.py --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("%.1f h2h %.1f lat %.2f" % (d["value"], d["host_to_host"]["value"], d["single_proof_latency_ms"]))'); echo "$*: $v"; }
python -m pytest tests/test_gpu_batch_mode.py tests/test_golden_proofs.py -m gpu -x -q 2>&1 | tail -2
run X=1
run MASP_HIP_BATCH=128
run MASP_HIP_BATCH=128 MASP_HIP_MSM_CHUNKS=8192
run X=1
run MASP_HIP_BATCH=128
run MASP_HIP_BATCH=128 MASP_HIP_MSM_CHUNKS=8192
