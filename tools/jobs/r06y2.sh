#!/bin/bash
# round 6: the driver's round-end sequence rehearsed on a fresh box — GPU tests, smoke(), bench.py at N = 1
o=gpurun_out/r06y2; mkdir -p $o
python -m pytest tests/ -x -q -m gpu > $o/gpu_tests.txt 2>&1; tail -3 $o/gpu_tests.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $o/smoke.txt 2>&1; tail -2 $o/smoke.txt
python bench.py --steps 20 --warmup 5 > $o/bench.json 2>> $o/bench.err; cut -c1-200 $o/bench.json
