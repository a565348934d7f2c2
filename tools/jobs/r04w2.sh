#!/bin/bash
# r04w2: the round's evidence in one call: GPU suite, default bench line (20 steps), PMC traffic, kernel statistics (one slot and default), SQ counters, lone-proof timeline
o=gpurun_out/r04w2; mkdir -p $o
timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > $o/tests.txt; cat $o/tests.txt
python bench.py --steps 20 --warmup 2 > $o/bench20.json 2> $o/bench20.err || tail -5 $o/bench20.err
python - <<PY
import json
d=json.loads(open("$o/bench20.json").read().strip().splitlines()[-1])
print("BENCH value %.1f resident %.1f e2e %s lat %.2f frac %.5f" % (d["value"], d["resident"]["value"], d["end_to_end"] and round(d["end_to_end"]["value"],1), d["single_proof_latency_ms"], d["roofline"]["frac"]))
print(d["roofline"].get("kernel_ms_per_launch"), d["roofline"]["avg_launch_ms"])
print(d["host_synthesis"]["instances_per_s_all_threads"], d.get("cpu_baseline",{}).get("phase_ms_per_proof"))
PY
PMC_STEPS=3 PMC_OUT=r04w2_pmc_traffic MASP_BENCH_E2E=0 bash tools/pmc_traffic.sh > $o/pmc.log 2>&1
PROF_ARGS="--steps 4 --warmup 1 --no-cpu-baseline" PROF_GY=256 MASP_BENCH_E2E=0 bash tools/prof_run.sh r04w2_slots1 MASP_HIP_SLOTS=1 > $o/prof_slots1.txt 2>&1
PROF_ARGS="--steps 8 --warmup 2 --no-cpu-baseline" PROF_GY=256 MASP_BENCH_E2E=0 bash tools/prof_run.sh r04w2_default > $o/prof_default.txt 2>&1
PMC_OUT=r04w2_pmc_sq bash tools/pmc_sq_kernels.sh > $o/pmc_sq.txt 2>&1
python tools/lone_timeline.py $(find gpurun_out/prof_r04w2_default -name "*.db" | head -1) > $o/lone_timeline.txt 2>&1
tail -3 $o/lone_timeline.txt
rm -f gpurun_out/prof_r04w2_default/*.db gpurun_out/prof_r04w2_slots1/*.db
