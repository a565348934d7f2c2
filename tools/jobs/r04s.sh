#!/bin/bash
o=gpurun_out/r04s; mkdir -p $o
timeout 900 python -m pytest tests/test_gpu_endomorphism.py -x -q 2>&1 | tail -15 > $o/endo.txt; cat $o/endo.txt
timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > $o/tests.txt; cat $o/tests.txt
