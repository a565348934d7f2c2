#!/bin/bash
# Where do the wave cycles of the big kernels go?  One rocprofv3 --pmc pass of 8 SQ counters over the default bench with ONE
# batch in flight (every kernel alone on the chip), summed per kernel (template arguments kept) over all of its dispatches.
# usage (on the GPU box): PMC_OUT=<name> tools/pmc_sq_kernels.sh   -> gpurun_out/<name>.json
export TMPDIR=/tmp
root=$PWD
out=$root/gpurun_out/pmc_sqk
rm -rf $out; mkdir -p $out
ctrs="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_WAVES"
(cd /tmp && MASP_HIP_SLOTS=1 MASP_BENCH_E2E=0 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $out -o run -- python $root/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $out/run.log 2>&1)
python - <<PY
import csv, glob, json, collections, re
f = glob.glob("$out/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.Counter()
for r in csv.DictReader(open(f)):
    k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("masp::", "")
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVES": n[k] += 1
doc = {"command": "tools/pmc_sq_kernels.sh: MASP_HIP_SLOTS=1 rocprofv3 --pmc <8 SQ counters> --kernel-trace -- python bench.py --steps 2 --warmup 1",
       "note": "one batch in flight; counters summed over all dispatches of a kernel; SQ_WAIT_* / SQ_ACTIVE_* in quad-cycles (MI355X_MICROARCH.md)", "kernels": {}}
tot = sum(v.get("SQ_BUSY_CYCLES", 0) for v in acc.values()) or 1
for k, v in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_BUSY_CYCLES", 0))[:28]:
    wc = v.get("SQ_ACTIVE_INST_ANY", 0) + v.get("SQ_WAIT_ANY", 0) + v.get("SQ_WAIT_INST_ANY", 0)
    doc["kernels"][k] = {"dispatches": n[k], "busy_share_of_all_kernels": round(v.get("SQ_BUSY_CYCLES", 0) / tot, 4),
                         "valu_issue_share": round(v.get("SQ_ACTIVE_INST_VALU", 0) / wc, 3) if wc else None,
                         "wait_any_share (memory, barriers)": round(v.get("SQ_WAIT_ANY", 0) / wc, 3) if wc else None,
                         "wait_inst_share (issue stalls)": round(v.get("SQ_WAIT_INST_ANY", 0) / wc, 3) if wc else None,
                         "valu_insts_per_wave": round(v.get("SQ_INSTS_VALU", 0) / v["SQ_WAVES"], 1) if v.get("SQ_WAVES") else None,
                         "vmem_reads_per_wave": round(v.get("SQ_INSTS_VMEM_RD", 0) / v["SQ_WAVES"], 1) if v.get("SQ_WAVES") else None}
json.dump(doc, open("$root/gpurun_out/${PMC_OUT:-pmc_sq_kernels}.json", "w"), indent=1)
for k, d in doc["kernels"].items():
    print("%-52s busy %.3f  valu %.2f  wait %.2f  stall %.2f  valu/wave %s" % (k[:52], d["busy_share_of_all_kernels"], d["valu_issue_share"] or 0, d["wait_any_share (memory, barriers)"] or 0, d["wait_inst_share (issue stalls)"] or 0, d["valu_insts_per_wave"]))
PY
