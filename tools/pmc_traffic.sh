#!/bin/bash
# HBM traffic of the dominant kernel (k_msm_accumulate<G1>) from the PMC counters, in two separate rocprofv3 passes
# (FETCH_SIZE and WRITE_SIZE do not fit one pass; --pmc is never combined with traces other than --kernel-trace).
# Writes profiles/pmc_traffic.json.  usage (on the GPU box): tools/pmc_traffic.sh
export TMPDIR=/tmp
root=$PWD
out=$root/gpurun_out/pmc
rm -rf $out; mkdir -p $out
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/$c -o run -- python $root/bench.py --steps 8 --warmup 2 --no-cpu-baseline > $out/$c.log 2>&1)
done
python - <<PY
import csv, glob, json, os
res = {}
batch = int(os.environ.get("MASP_HIP_BATCH", "96"))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob("$out/%s/**/*counter_collection.csv" % c, recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if "k_msm_accumulate<masp::FpOps>" in r["Kernel_Name"] and r["Counter_Name"] == c]
    full = max(int(r["Grid_Size"]) for r in rows)          # the full batches (lone-proof launches have another grid)
    vals = [float(r["Counter_Value"]) for r in rows if int(r["Grid_Size"]) == full]
    res[c] = (sum(vals) / len(vals), len(vals))
fetch_kb, n1 = res["FETCH_SIZE"]; write_kb, n2 = res["WRITE_SIZE"]
json.dump({
 "kernel": "k_msm_accumulate<G1>",
 "command": "tools/pmc_traffic.sh: rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (separate passes) --kernel-trace -- python bench.py --steps 8 --warmup 2 --no-cpu-baseline",
 "FETCH_SIZE_KB_avg_per_launch": fetch_kb, "WRITE_SIZE_KB_avg_per_launch": write_kb, "launches_sampled": min(n1, n2),
 "hbm_bytes_per_launch": (fetch_kb + write_kb) * 1024.0, "proofs_per_launch": batch,
 "note": "one launch covers one G1 query (h, l, a or b_g1) of a batch of %d proofs; counters averaged over the four queries. FETCH_SIZE is used as reported: the gfx950 x2 correction of MI355X_MICROARCH.md is calibrated for wide coalesced streams only, these are 96-byte gathers of window-table rows (doubling it gives the upper bound). Traffic exceeds the algorithmic bytes (n x 128 B per proof) because every non-zero digit reads its own 96-byte table row: that is the HBM-capacity-for-ALU trade of DESIGN.md." % batch,
}, open("$root/profiles/pmc_traffic.json", "w"), indent=1)
print(open("$root/profiles/pmc_traffic.json").read())
PY
