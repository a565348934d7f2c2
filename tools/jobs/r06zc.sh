#!/bin/bash
# round 6, the end: the driver's round-end sequence once more on a fresh box after the C++ mirror and its tests were added
o=gpurun_out/r06zc; mkdir -p $o
python -m pytest tests/ -x -q -m gpu > $o/gpu_tests.txt 2>&1; tail -3 $o/gpu_tests.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $o/smoke.txt 2>&1; tail -2 $o/smoke.txt
python bench.py --steps 20 --warmup 5 > $o/bench.json 2>> $o/bench.err; cut -c1-200 $o/bench.json
