#!/bin/bash
# r04d: full GPU suite + default bench (end to end with the lockstep host synthesizer) + PMC traffic + one-slot kernel stats
o=gpurun_out/r04d; mkdir -p $o
timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > $o/tests.txt
cat $o/tests.txt
python bench.py > $o/bench.json 2> $o/bench.err || tail -5 $o/bench.err
python - <<PY
import json
d=json.loads(open("$o/bench.json").read().strip().splitlines()[-1])
print("BENCH value %.1f resident %.1f e2e %s lat %.2f host %s cpu %s" % (d["value"], d["resident"]["value"], d["end_to_end"] and round(d["end_to_end"]["value"],1), d["single_proof_latency_ms"], d["host_synthesis"], d.get("cpu_baseline",{}).get("value")))
PY
PMC_STEPS=2 PMC_OUT=r04d_pmc_traffic MASP_BENCH_E2E=0 bash tools/pmc_traffic.sh > $o/pmc.log 2>&1
PROF_ARGS="--steps 3 --warmup 1 --no-cpu-baseline" PROF_GY=256 MASP_BENCH_E2E=0 bash tools/prof_run.sh r04d_slots1 MASP_HIP_SLOTS=1 > $o/prof_slots1.txt 2>&1
head -16 gpurun_out/prof_r04d_slots1/all.txt | cut -c1-140
