#!/bin/bash
o=gpurun_out/r04e; mkdir -p $o
timeout 900 python -m pytest tests/test_gpu_bucket_tree.py tests/test_gpu_bench_dist.py -m gpu -x -q 2>&1 | tail -4 > $o/tests.txt; cat $o/tests.txt
timeout 1200 python tools/e2e_sweep.py 1024 > $o/e2e_sweep.txt 2>&1; cat $o/e2e_sweep.txt
