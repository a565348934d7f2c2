#!/bin/bash
# Window widths of the other circuits: usage tools/sweep_windows.sh <output|convert> "<c_h values>" "<c_la values>" "<c_b values>"
kind=$1
for ch in $2; do for cla in $3; do for cb in $4; do
  v=$(MASP_BENCH_CIRCUIT=$kind MASP_BENCH_LONE=0 MASP_BENCH_OTHER=0 MASP_BENCH_E2E=0 MASP_HIP_MSM_C_H=$ch MASP_HIP_MSM_C_LA=$cla MASP_HIP_MSM_C_B=$cb python bench.py --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.1f resident %.1f' % (d['value'], d['resident']['value']))")
  echo "$kind c_h=$ch c_la=$cla c_b=$cb: $v"
done; done; done
