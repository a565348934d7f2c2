#!/bin/bash
# VALU instructions of ONE batch of 256 Spend proofs and what they cost to issue: the numbers behind bench.py's `roofline_valu`.
#   usage (GPU box): tools/valu_model.sh [out.json = gpurun_out/valu_model.json]      then copy to profiles/r06_valu_model.json
# One rocprofv3 --pmc pass (SQ_INSTS_VALU: wave-level VALU instructions, summed per kernel) over bench.py with ONE slot and nothing but
# full batches in the run (no lone proofs, no other circuits, no end-to-end region); the kernels of set-up and verification are left out.
# Batches in the run = dispatches of k_tree_pass2<FpOps, true> / 9 (three G1 MSMs per batch, each through the tree in three sub-batches).
# Cycles per instruction: profiles/r06_static_valu_mix.json (tools/valu_mix.py: the class mix of each kernel's hot loop, priced with the
# measured issue costs), weighted with the counts.
export TMPDIR=/tmp
root=$PWD
out=$root/gpurun_out/pmc_valu
rm -rf $out; mkdir -p $out
(cd /tmp && MASP_HIP_SLOTS=1 MASP_BENCH_E2E=0 MASP_BENCH_OTHER=0 MASP_BENCH_LONE=0 rocprofv3 --pmc SQ_INSTS_VALU --kernel-trace --output-format csv -d $out -o run -- python $root/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $out/run.log 2>&1)
python - <<PY
import csv, glob, json, collections, re
f = glob.glob("$out/**/*counter_collection.csv", recursive=True)[0]
insts, disp = collections.defaultdict(float), collections.defaultdict(int)
for r in csv.DictReader(open(f)):
    if r["Counter_Name"] != "SQ_INSTS_VALU":
        continue
    k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("masp::", "")
    insts[k] += float(r["Counter_Value"])
    disp[k] += 1
NOT_A_BATCH = ("k_setup", "k_fixed_table", "k_msm_precompute", "k_msm_import", "k_verify", "k_miller", "k_fp12", "k_g1_sum_export", "k_fr_powers", "k_g1_subgroup",
               "k_subgroup", "k_msm_table", "k_msm_window")
batches = disp["k_tree_pass2<FpOps, true>"] / 9.0
mix = json.load(open("$root/profiles/${VALU_MIX:-r06_static_valu_mix.json}"))
kern = mix["kernels"]
rows, tot, cyc, left_out = [], 0.0, 0.0, {}
for k, v in sorted(insts.items(), key=lambda kv: -kv[1]):
    if k.startswith(NOT_A_BATCH) or not k.startswith("k_"):
        left_out[k] = v
        continue
    cpi = (kern.get(k) or {}).get("cycles_per_valu_instruction")
    # kernels whose products are out-of-line calls (the G2 tails): their loop's mix misses the callee, which is all multiply-adds and
    # carry words — priced like the inline product kernels
    if cpi is None or (kern[k]["classes"].get("mad64", 0) < 0.2 * kern[k]["valu"]):
        cpi = 4.3
    tot += v
    cyc += v * cpi
    rows.append({"kernel": k, "valu_insts_per_batch": v / batches, "dispatches_per_batch": disp[k] / batches, "cycles_per_inst": cpi})
import hashlib
doc = {"library_sha16": hashlib.sha256(open("$root/masp_amd/libmasp_hip.so", "rb").read()).hexdigest()[:16],
       "what": "wave-level VALU instructions of one batch of 256 Spend proofs (rocprofv3 --pmc SQ_INSTS_VALU over bench.py, one slot) and their issue cost",
       "batches_in_run": batches, "valu_insts_per_batch": tot / batches, "cycles_per_inst_weighted": cyc / tot, "simds": 1024,
       "issue_cycles_per_batch_per_simd": cyc / batches / 1024, "kernels": rows[:24],
       "left_out_not_part_of_a_batch": {k: v for k, v in sorted(left_out.items(), key=lambda kv: -kv[1])[:12]},
       "cost_source": mix["source"], "mix_source": "profiles/${VALU_MIX:-r06_static_valu_mix.json}"}
json.dump(doc, open("$root/${1:-gpurun_out/valu_model.json}", "w"), indent=1)
print(json.dumps({k: doc[k] for k in ("batches_in_run", "valu_insts_per_batch", "cycles_per_inst_weighted", "issue_cycles_per_batch_per_simd")}))
for r in rows[:12]: print(r)
print("left out:", list(doc["left_out_not_part_of_a_batch"])[:12])
PY
