#!/bin/bash
# long runs and the other workloads on the round's final build (one box): bench --steps 64, Output / Convert / mixed, 4 096 mixed descriptions end to end
o=gpurun_out/r04k; mkdir -p $o
sel='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); e=d.get("end_to_end") or {}; print("%s: value %.1f  resident %.1f  end_to_end %s  lone %.2f ms  (%d steps, %.1f ms per step)" % (d["config"]["workload"][:60], d["value"], d["resident"]["value"], ("%.1f" % e["value"]) if e else "-", d["single_proof_latency_ms"], d["steps"], d["ms_per_step"]))'
python bench.py --steps 64 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "$sel" > $o/long_runs.txt
for w in output convert mixed; do MASP_BENCH_CIRCUIT=$w python bench.py --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "$sel" >> $o/other_workloads.txt; done
python tools/soak_mixed.py 4096 >> $o/long_runs.txt 2>&1
cat $o/long_runs.txt $o/other_workloads.txt
