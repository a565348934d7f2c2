#!/bin/bash
# Where do the wave cycles of the dominant kernel go?  One rocprofv3 --pmc pass of SQ counters (8 slots) with the kernel
# running alone (one slot), summarised per launch of k_msm_accumulate<G1> into profiles/pmc_sq_accumulate.json.
export TMPDIR=/tmp
root=$PWD
out=$root/gpurun_out/pmc_sq
rm -rf $out; mkdir -p $out
ctrs="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_WAVES"
(cd /tmp && MASP_HIP_SLOTS=1 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $out -o run -- python $root/bench.py --steps 4 --warmup 1 --no-cpu-baseline > $out/run.log 2>&1)
python - <<PY
import csv, glob, json, collections
f = glob.glob("$out/**/*counter_collection.csv", recursive=True)[0]
allrows = list(csv.DictReader(open(f)))
doc = {"command": "tools/pmc_sq.sh: MASP_HIP_SLOTS=1 rocprofv3 --pmc <8 SQ counters> --kernel-trace -- python bench.py --steps 4 --warmup 1 --no-cpu-baseline",
       "note": "one batch in flight (every kernel alone on the chip), averages per launch of the full 128-proof batch; quad-cycle units (MI355X_MICROARCH.md)"}
for key, pat, vgprs in (("k_msm_accumulate<G1>", "k_msm_accumulate<masp::FpOps>", "226 VGPRs, no spills: 2 waves per SIMD"),
                        ("k_msm_accumulate<G2>", "k_msm_accumulate<masp::Fp2Ops>", "396 VGPRs, no spills in the loop (the exceptional cases of the addition are one out-of-line call: 512 B of scratch for its arguments): 1 wave per SIMD; the 384-bit product is an out-of-line call")):
    rows = [r for r in allrows if pat in r["Kernel_Name"]]
    full = max(int(r["Grid_Size"]) for r in rows)
    acc = collections.defaultdict(list)
    for r in rows:
        if int(r["Grid_Size"]) == full:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    avg = {k: sum(v) / len(v) for k, v in acc.items()}
    wave_cycles = avg.get("SQ_ACTIVE_INST_ANY", 0) + avg.get("SQ_WAIT_ANY", 0) + avg.get("SQ_WAIT_INST_ANY", 0)
    doc[key] = {"registers": vgprs, "launches_sampled": len(next(iter(acc.values()))), "counters": avg,
                "derived": {"wave_quad_cycles (ACTIVE_INST_ANY + WAIT_ANY + WAIT_INST_ANY)": wave_cycles,
                            "share issuing VALU": avg.get("SQ_ACTIVE_INST_VALU", 0) / wave_cycles if wave_cycles else None,
                            "share waiting on memory / barriers (WAIT_ANY)": avg.get("SQ_WAIT_ANY", 0) / wave_cycles if wave_cycles else None,
                            "share stalled at issue (WAIT_INST_ANY)": avg.get("SQ_WAIT_INST_ANY", 0) / wave_cycles if wave_cycles else None,
                            "VALU instructions per wave": avg.get("SQ_INSTS_VALU", 0) / avg["SQ_WAVES"] if avg.get("SQ_WAVES") else None}}
json.dump(doc, open("$root/profiles/pmc_sq_accumulate.json", "w"), indent=1)
print(json.dumps(doc, indent=1))
PY
cp $root/profiles/pmc_sq_accumulate.json $root/gpurun_out/pmc_sq_accumulate.json   # gpurun only merges gpurun_out/ back: copy it into profiles/ afterwards
