#!/bin/bash
o=gpurun_out/r04g; mkdir -p $o
timeout 2400 python -m pytest tests -m gpu -x -q --durations=8 2>&1 | tail -16 > $o/tests.txt; cat $o/tests.txt
