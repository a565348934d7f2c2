#!/bin/bash
o=gpurun_out/r04n; mkdir -p $o
timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > $o/tests.txt; cat $o/tests.txt
MASP_BENCH_E2E=0 bash tools/ab.sh masp_amd/libmasp_hip_B.so masp_amd/libmasp_hip.so 2 > $o/ab.txt 2>&1; cat $o/ab.txt
PMC_STEPS=3 PMC_OUT=r04n_pmc_traffic MASP_BENCH_E2E=0 bash tools/pmc_traffic.sh > $o/pmc.log 2>&1
python -c "
import json; d=json.load(open('gpurun_out/r04n_pmc_traffic.json')); print('traffic GB per launch', d['hbm_bytes_per_launch']/1e9)"
