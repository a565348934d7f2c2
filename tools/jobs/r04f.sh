#!/bin/bash
o=gpurun_out/r04f; mkdir -p $o
timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > $o/tests.txt; cat $o/tests.txt
python bench.py > $o/bench.json 2> $o/bench.err || tail -5 $o/bench.err
python - <<PY
import json
d=json.loads(open("$o/bench.json").read().strip().splitlines()[-1])
print("BENCH value %.1f resident %.1f e2e %s lat %.2f" % (d["value"], d["resident"]["value"], d["end_to_end"] and round(d["end_to_end"]["value"],1), d["single_proof_latency_ms"]))
print(d["host_synthesis"])
PY
