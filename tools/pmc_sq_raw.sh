#!/bin/bash
# Raw SQ counter sums per kernel (one batch in flight): usage (GPU box): tools/pmc_sq_raw.sh "<counters>" <out.json>
export TMPDIR=/tmp
root=$PWD
out=$root/gpurun_out/pmc_sqr
rm -rf $out; mkdir -p $out
ctrs=${1:-"SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES"}
(cd /tmp && MASP_HIP_SLOTS=1 MASP_BENCH_E2E=0 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $out -o run -- python $root/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $out/run.log 2>&1)
python - <<PY
import csv, glob, json, collections, re
f = glob.glob("$out/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
dur = collections.defaultdict(float)
for r in csv.DictReader(open(f)):
    k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("masp::", "")
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"].endswith("SQ_WAVES"):
        acc[k]["dispatches"] += 1
        if "End_Timestamp" in r: dur[k] += (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) * 1e-3
doc = {k: dict(v, total_us=round(dur[k], 1)) for k, v in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_BUSY_CYCLES", kv[1].get("SQ_WAVES", 0)))[:16]}
json.dump(doc, open("$root/gpurun_out/${2:-pmc_sq_raw.json}", "w"), indent=1)
print(json.dumps({k: doc[k] for k in list(doc)[:6]}, indent=1))
PY
