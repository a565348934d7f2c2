#!/usr/bin/env python3
"""Static instruction mix of a kernel's hot loop from the device assembly (make -C masp_amd/csrc _build/<unit>.s).

    tools/asm_loop_stats.py <file.s> <kernel-name substring> [--all]

Finds the kernel's body, its basic blocks and the backward branches; reports the opcode histogram of the whole kernel and of
every loop (label .. backward branch), largest first.  Classes follow the measured issue costs of
profiles/r04e_valu_instruction_cost_classes_ubench.txt: plain VOP1/VOP2 (2.45 cycles per wave64 instruction), VOP3 / carry /
64-bit multiply-add (4.4 - 4.9), v_mul_lo/hi_u32 and 64-bit shifts (~12)."""
import collections
import re
import sys


def klass(op):
    if not op.startswith("v_"):
        return "salu/other" if op.startswith("s_") else ("vmem" if op.startswith(("global_", "buffer_", "flat_", "scratch_")) else ("lds" if op.startswith("ds_") else "other"))
    if op.startswith(("v_mad_u64_u32", "v_mad_i64_i32")):
        return "mad64"
    if op.startswith(("v_mul_lo_u32", "v_mul_hi_u32", "v_lshrrev_b64", "v_lshlrev_b64", "v_ashrrev_i64")):
        return "slow12"
    if re.match(r"v_(addc|subb|subbrev)_co|v_(add|sub|subrev)_co", op):
        return "carry"
    if op.startswith("v_mov_b32"):
        return "mov"
    if op.startswith(("v_accvgpr",)):
        return "accvgpr"
    if op.endswith("_e64") or op.startswith(("v_cndmask", "v_add3", "v_lshl_add", "v_alignbit", "v_mad_", "v_and_or", "v_or3", "v_xad", "v_bfe", "v_perm", "v_cmp", "v_readlane", "v_readfirstlane", "v_writelane", "v_lshl_or", "v_add_lshl")):
        return "vop3/cmp"
    return "vop2/1"


COST = {"mad64": 4.8, "carry": 4.5, "vop3/cmp": 4.5, "slow12": 12.0, "mov": 2.45, "vop2/1": 2.45, "accvgpr": 2.45}


def main():
    path, name = sys.argv[1], sys.argv[2]
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and name in l and l.rstrip().endswith(tuple(": ; @" + l.split(":")[0] for _ in [0])) or (l.startswith("_Z") and name in l.split(":")[0] and ":" in l))
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    labels, insts = {}, []
    for i in range(start + 1, end + 1):
        l = lines[i].split(";")[0].strip()
        if not l or l.startswith("."):
            m = re.match(r"(\.LBB\d+_\d+):", l)
            if m:
                labels[m.group(1)] = len(insts)
            continue
        m = re.match(r"(\.LBB\d+_\d+):", l)
        if m:
            labels[m.group(1)] = len(insts)
            continue
        insts.append(l)
    ops = [l.split()[0] for l in insts]
    tot = collections.Counter(klass(o) for o in ops)
    print("kernel %s: %d instructions" % (lines[start].split(":")[0][:70], len(insts)), dict(tot))
    loops = []
    for k, l in enumerate(insts):
        m = re.match(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)|s_branch\s+(\.LBB\d+_\d+)", l)
        if m:
            tgt = labels.get(m.group(1) or m.group(2))
            if tgt is not None and tgt <= k:
                loops.append((tgt, k))
    loops.sort(key=lambda t: t[0] - t[1])
    for lo, hi in loops[: (len(loops) if "--all" in sys.argv else 3)]:
        c = collections.Counter(klass(o) for o in ops[lo:hi + 1])
        valu = sum(v for k, v in c.items() if k in COST)
        cyc = sum(COST[k] * v for k, v in c.items() if k in COST)
        opsc = collections.Counter(o for o in ops[lo:hi + 1] if klass(o) in ("vop3/cmp", "vop2/1", "slow12"))
        print("  loop [%d, %d]: %d instructions, VALU %d, priced %.0f cycles (%.2f per VALU instruction)" % (lo, hi, hi - lo + 1, valu, cyc, cyc / max(valu, 1)))
        print("    classes:", dict(c))
        print("    other VALU opcodes:", dict(opsc.most_common(14)))


if __name__ == "__main__":
    main()
