#!/usr/bin/env python3
"""Static VALU instruction mix of every kernel of libmasp_hip.so, priced with the measured issue costs per class
(profiles/r06_valu_instruction_cost_classes_ubench.txt, first measured in round 4: a wave64 instruction occupies its SIMD for ~2.45 cycles if it is a plain
VOP1 / VOP2, ~4.5 if it is VOP3-encoded, reads or writes a carry, ~4.8 for v_mad_u64_u32, ~12 for v_mul_lo/hi_u32 and 64-bit shifts).

    make -C masp_amd/csrc asm && tools/valu_mix.py masp_amd/csrc/_build/*.s > profiles/r06_static_valu_mix.json

Per kernel: the class histogram of its hottest loop (the longest backward-branch span; the whole body if it has no loop) and the
cycles per VALU instruction that follow.  tools/valu_model.sh weights these with the SQ_INSTS_VALU counts of a bench run."""
import collections
import json
import re
import subprocess
import sys

sys.path.insert(0, __import__("os").path.dirname(__file__))
from asm_loop_stats import COST, klass  # noqa: E402


def kernels(path):
    lines = open(path).read().split("\n")
    names = [m.group(1) for m in (re.match(r"\s*\.amdhsa_kernel\s+(\S+)", l) for l in lines) if m]
    for name in names:
        try:
            start = next(i for i, l in enumerate(lines) if l.startswith(name + ":"))
        except StopIteration:
            continue
        end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
        labels, insts = {}, []
        for i in range(start + 1, end + 1):
            l = lines[i].split(";")[0].strip()
            m = re.match(r"(\.LBB\d+_\d+):", l)
            if m:
                labels[m.group(1)] = len(insts)
                continue
            if not l or l.startswith("."):
                continue
            insts.append(l)
        ops = [l.split()[0] for l in insts]
        lo, hi = 0, len(ops) - 1
        best = 0
        for k, l in enumerate(insts):
            m = re.match(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)|s_branch\s+(\.LBB\d+_\d+)", l)
            if m:
                tgt = labels.get(m.group(1) or m.group(2))
                if tgt is not None and tgt <= k and k - tgt > best:
                    best, lo, hi = k - tgt, tgt, k
        c = collections.Counter(klass(o) for o in ops[lo:hi + 1])
        valu = sum(v for kk, v in c.items() if kk in COST)
        cyc = sum(COST[kk] * v for kk, v in c.items() if kk in COST)
        yield name, {"loop_instructions": hi - lo + 1, "valu": valu, "classes": dict(c), "cycles_per_valu_instruction": round(cyc / valu, 3) if valu else None}


def main():
    out = {}
    for path in sys.argv[1:]:
        for name, rec in kernels(path):
            out[name] = rec
    dem = subprocess.run(["c++filt"], input="\n".join(out), capture_output=True, text=True).stdout.split("\n")
    res = {}
    for (name, rec), d in zip(out.items(), dem):
        short = re.sub(r"\(.*", "", d).replace("void ", "").replace("masp::", "")
        rec["mangled"] = name
        res[short] = rec
    json.dump({"cost_cycles_per_class": COST, "source": "profiles/r06_valu_instruction_cost_classes_ubench.txt (tools/valu_rate_ubench.hip on the round's final box; the classes' costs are those of round 4's measurement)", "kernels": res}, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
