#!/bin/bash
# HBM traffic of the dominant stage (G1 bucket accumulation: batch-affine tree + k_msm_accumulate_pts<G1>) from the PMC counters, as MI355X_MICROARCH.md (§HBM, §rocprofv3
# PMC slots) prescribes: FETCH_SIZE and WRITE_SIZE in SEPARATE rocprofv3 passes (they do not fit one), --pmc combined with
# nothing but --kernel-trace, and FETCH_SIZE calibrated on a known byte count in THIS access pattern (tools/pmc_calib.hip:
# one 96-byte row gathered per lane at a 128-byte stride from a 3 GiB table) instead of assuming the x2 of wide streams.
# The bench runs its default workload: 256 DISTINCT Spend witnesses per step — and nothing but Spend batches (MASP_BENCH_OTHER=0: since round 5 the bench also
# proves Output / Convert batches of 256, whose G1 MSMs would be counted as stages and dilute the per-MSM figure; the end-to-end region proves Spend batches).  Writes profiles/pmc_traffic.json.
# usage (on the GPU box): tools/pmc_traffic.sh
export TMPDIR=/tmp
root=$PWD
out=$root/gpurun_out/pmc
rm -rf $out; mkdir -p $out
[ -x $root/tools/_build/pmc_calib ] || { mkdir -p $root/tools/_build; /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 $root/tools/pmc_calib.hip -o $root/tools/_build/pmc_calib; }
(cd /tmp && rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/calib -o run -- $root/tools/_build/pmc_calib > $out/calib.log 2>&1)
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && MASP_BENCH_OTHER=0 MASP_BENCH_LONE=0 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/$c -o run -- python $root/bench.py --steps ${PMC_STEPS:-4} --warmup 1 --no-cpu-baseline > $out/$c.log 2>&1)
done
python - <<PY
import csv, glob, json, os, re
out = "$out"
def rows_of(d, counter):
    f = glob.glob("%s/%s/**/*counter_collection.csv" % (out, d), recursive=True)[0]
    return [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == counter]
# ---- calibration: reported KB per launch vs the bytes of the 128-byte lines the gather touches
cal = rows_of("calib", "FETCH_SIZE")
rows_per_launch = (1 << 20) * 16
def avg(rows, pat):
    v = [float(r["Counter_Value"]) for r in rows if pat in r["Kernel_Name"]]
    return sum(v) / len(v)
rep128, rep96 = avg(cal, "k_calib_gather<128>"), avg(cal, "k_calib_gather<96>")
true128 = rows_per_launch * 128.0          # one line per row
true96 = rows_per_launch * 1.5 * 128.0     # a 96-byte row at a 96-byte stride touches 1.5 lines on average
factor = true128 / (rep128 * 1024.0)
# ---- the G1 bucket-accumulation stage: since round 3 a group of kernels per G1 MSM (batch-affine tree levels in sub-batches of
# 64 proofs, then the XYZZ accumulation of what is left).  Counters summed over every dispatch of the stage's kernels, divided by
# the number of G1 MSMs of full batches (= dispatches of k_msm_combine<G1> with gridDim.y = batch: one per MSM).
res = {}
batch = int(os.environ.get("MASP_HIP_BATCH", "256"))
# (k_tree_plan / k_tree_records are curve-independent kernels: their G2 dispatches — a quarter of them, ~1 % of the stage's bytes —
# are counted too)
STAGE = ("k_tree_pass1<masp::FpOps", "k_tree_pass2<masp::FpOps", "k_tree_copy<masp::FpOps", "k_binv_fwd<masp::FpOps", "k_binv_mid<masp::FpOps",
         "k_binv_bwd<masp::FpOps", "k_msm_accumulate_pts<masp::FpOps", "k_msm_accumulate<masp::FpOps", "k_tree_records", "k_tree_plan")
per_kernel = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    rows = rows_of(c, c)
    msms = sum(1 for r in rows if "k_msm_combine<masp::FpOps>" in r["Kernel_Name"] and int(r["Grid_Size"]) == 64 * batch)
    tot = 0.0
    for r in rows:
        for k in STAGE:
            # (k_msm_accumulate<G1> itself only runs for lone proofs now: gridDim.y = 1, excluded)
            if k in r["Kernel_Name"] and not ("k_msm_accumulate<masp::FpOps" in k and int(r["Grid_Size"]) < 64 * 3072):
                tot += float(r["Counter_Value"])
                kk = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("masp::", "")   # template arguments kept: level 0 / deeper levels apart
                per_kernel.setdefault(kk, {}).setdefault(c, 0.0)
                per_kernel[kk][c] += float(r["Counter_Value"])
    res[c] = (tot / max(msms, 1), msms)
fetch_kb, n1 = res["FETCH_SIZE"]; write_kb, n2 = res["WRITE_SIZE"]
for k in per_kernel:
    for c in per_kernel[k]:
        per_kernel[k][c] = per_kernel[k][c] * 1024.0 / max(n1, 1)     # bytes per G1 MSM, as reported (uncalibrated)
# algorithmic bytes per launch as the bench itself reports them (n x 128 B per G1 MSM and proof; a batch runs three G1
# accumulations: H+L merged, A, B1)
import re, time
line = [l for l in open("%s/FETCH_SIZE.log" % out).read().splitlines() if l.startswith("{")][-1]
alg = json.loads(line)["roofline"]["alg_bytes_per_launch"]
import hashlib
doc = {
 "library_sha16": hashlib.sha256(open("$root/masp_amd/libmasp_hip.so", "rb").read()).hexdigest()[:16],
 "kernel": "G1 bucket-accumulation stage per G1 MSM: k_tree_plan / k_tree_records / k_tree_pass1 / k_tree_pass2 / k_tree_copy / k_binv_* over 4 tree levels in sub-batches of 86 proofs, then k_msm_accumulate_pts",
 "date": time.strftime("%Y-%m-%d"),
 "command": "tools/pmc_traffic.sh: rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (separate passes) --kernel-trace -- python bench.py --steps 4 --warmup 1 --no-cpu-baseline (256 distinct Spend witnesses per step); calibration pass on tools/_build/pmc_calib",
 "calibration": {"pattern": "one 96-byte row (6 x global_load_dwordx4) per lane at a random index, 3 GiB table, 2^24 rows per launch",
                 "stride128_reported_KB": rep128, "stride128_line_bytes": true128, "stride96_reported_KB": rep96, "stride96_expected_line_bytes": true96,
                 "bytes_per_reported_byte": factor, "stride96_over_stride128_reported": rep96 / rep128,
                 "reading": "FETCH_SIZE x %.3f = bytes of 128-B lines fetched for this gather pattern (the guide's x2 holds for wide coalesced streams)" % factor},
 "FETCH_SIZE_KB_per_g1_msm": fetch_kb, "WRITE_SIZE_KB_per_g1_msm": write_kb, "g1_msms_sampled": min(n1, n2),
 "fetch_bytes_per_launch_calibrated": fetch_kb * 1024.0 * factor, "write_bytes_per_launch_as_reported": write_kb * 1024.0,
 "hbm_bytes_per_launch": fetch_kb * 1024.0 * factor + write_kb * 1024.0, "proofs_per_launch": batch,
 "reported_bytes_per_g1_msm_by_kernel": per_kernel,
 "algorithmic_bytes_per_launch": alg, "traffic_over_algorithmic": (fetch_kb * 1024.0 * factor + write_kb * 1024.0) / alg,
 "note": "one 'launch' = the bucket accumulation of one G1 MSM (h+l merged, a, or b_g1) of a batch of %d proofs; counters summed over the stage's kernels and averaged over the three MSMs. Traffic exceeds the algorithmic bytes (n x 128 B per proof) because (1) every non-zero window digit reads its own 128-byte table row (16 rows per full-width scalar of h and l, 22 per non-trivial scalar of a / b_g1) - the HBM-capacity-for-ALU trade of DESIGN.md - and (2) the shared-inversion tree reads every point of a level twice (denominators, then additions) and keeps 48 bytes per pair in between: memory traffic bought to save 40 %% of the field products. WRITE_SIZE is uncalibrated." % batch,
}
json.dump(doc, open("$root/gpurun_out/${PMC_OUT:-pmc_traffic}.json", "w"), indent=1)   # gpurun only merges gpurun_out/ back: copy it into profiles/ afterwards
print(json.dumps(doc, indent=1))
PY
