#!/bin/bash
python -m pytest tests -m gpu -x -q 2>&1 | tail -2
bash tools/ab.sh masp_amd/libmasp_hip_base.so masp_amd/libmasp_hip.so 2
run() { v=$(env "$@" python bench.py --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("%.1f h2h %.1f lat %.2f" % (d["value"], d["host_to_host"]["value"], d["single_proof_latency_ms"]))'); echo "$*: $v"; }
run MASP_HIP_NTT_SUB=10
run MASP_HIP_NTT_SUB=12
run MASP_HIP_NTT_SUB=8
