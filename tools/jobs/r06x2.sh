#!/bin/bash
# round 6, final binary: the mixed workload's line again after bench.py's sizing-steps fix (every slot sees every circuit before timing), and the default line on one more box
o=gpurun_out/r06x2; mkdir -p $o
MASP_BENCH_CIRCUIT=mixed python bench.py --steps 8 --warmup 2 --no-cpu-baseline > $o/bench_mixed_workload.json 2>> $o/bench.err; cut -c1-160 $o/bench_mixed_workload.json
python bench.py --steps 20 --warmup 5 > $o/bench_driver_flags_steps20_warmup5.json 2>> $o/bench.err; cut -c1-200 $o/bench_driver_flags_steps20_warmup5.json
python bench.py > $o/bench_no_flags.json 2>> $o/bench.err; cut -c1-200 $o/bench_no_flags.json
sha256sum masp_amd/libmasp_hip.so | cut -c1-16 > $o/library_sha16.txt
