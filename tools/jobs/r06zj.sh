#!/bin/bash
# round 6, the end: the driver's round-end sequence on a fresh box with the final tree
o=gpurun_out/r06zj; mkdir -p $o
python -m pytest tests/ -x -q -m gpu > $o/gpu_tests.txt 2>&1; tail -3 $o/gpu_tests.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $o/smoke.txt 2>&1; tail -2 $o/smoke.txt
python bench.py --steps 20 --warmup 5 > $o/bench.json 2>> $o/bench.err; cut -c1-200 $o/bench.json
python bench.py > $o/bench_no_flags.json 2>> $o/bench.err; cut -c1-120 $o/bench_no_flags.json
sha256sum masp_amd/libmasp_hip.so | cut -c1-16
