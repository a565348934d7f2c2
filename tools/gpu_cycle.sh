#!/bin/bash
# One GPU round trip of a kernel-tuning cycle: parity tests, the default bench line, and two rocprofv3 kernel traces of the
# bench (default = 4 slots overlapped; MASP_HIP_SLOTS=1 = every kernel alone on the chip).  usage: tools/gpu_cycle.sh <tag> [quick]
tag=$1
mkdir -p gpurun_out/$tag
if [ "$2" != "quick" ]; then
  python -m pytest tests -m gpu -x -q 2>&1 | tail -3
else
  python -m pytest tests/test_gpu_batch_mode.py -m gpu -x -q 2>&1 | tail -3
fi
python bench.py --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/$tag/bench.json 2> gpurun_out/$tag/bench.err || tail -5 gpurun_out/$tag/bench.err
python - <<PY
import json
d=json.loads(open("gpurun_out/$tag/bench.json").read().strip().splitlines()[-1])
print("BENCH value %.1f  resident %.1f  verified %d  latency %.2f ms  acc_launch %.2f ms" % (d["value"], d["resident"]["value"], d["verified"], d["single_proof_latency_ms"], d["roofline"]["avg_launch_ms"]))
PY
PROF_ARGS="--steps 3 --warmup 1 --no-cpu-baseline" PROF_GY=256 bash tools/prof_run.sh ${tag}_default > gpurun_out/$tag/prof_default.txt 2>&1
PROF_ARGS="--steps 2 --warmup 1 --no-cpu-baseline" PROF_GY=256 bash tools/prof_run.sh ${tag}_slots1 MASP_HIP_SLOTS=1 > gpurun_out/$tag/prof_slots1.txt 2>&1
grep -v "^W2026\|^E2026" gpurun_out/$tag/prof_slots1.txt | head -32
