#!/usr/bin/env python3
"""The G1 bucket-accumulation STAGE per Spend MSM out of a rocprofv3 kernel trace (rocpd SQLite database) — the table bench.py's
`roofline.avg_launch_ms` can be set against by somebody who was not there.

usage: g1_stage_stats.py results.db [> profiles/r06_g1_stage_per_spend_msm.txt]

The trace must come from a bench run that proves nothing but full Spend batches (tools/prof_run.sh with MASP_HIP_SLOTS=1
MASP_BENCH_OTHER=0 MASP_BENCH_E2E=0 MASP_BENCH_LONE=0: one batch in flight, so kernel durations do not overlap).  A stage = the
bucket accumulation of ONE G1 MSM of a 256-proof batch (h + l merged, a, or b_g1): the tree's kernels over FpOps in three sub-batches
of 86 / 85 / 85 proofs, then k_msm_accumulate_pts<FpOps>.  Stages are counted as dispatches of k_msm_combine<FpOps> with grid.y = 256
(one per G1 MSM).  k_tree_plan / k_tree_records are curve-independent kernels: a dispatch belongs to the curve of the next
k_tree_pass1 that follows it on the (single) stream.  The five groups are bench.py's `roofline.kernel_ms_per_launch`."""
import re
import sqlite3
import sys

GROUPS = (("plan_records_copies", ("k_tree_plan", "k_tree_records", "k_tree_copy")), ("k_tree_pass1", ("k_tree_pass1",)),
          ("k_binv", ("k_binv_fwd", "k_binv_mid", "k_binv_bwd")), ("k_tree_pass2", ("k_tree_pass2",)),
          ("k_msm_accumulate_pts", ("k_msm_accumulate_pts",)))


def main(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    sym_cols = [r[1] for r in cur.execute("pragma table_info(rocpd_info_kernel_symbol)")]
    name_col = "display_name" if "display_name" in sym_cols else "kernel_name"
    rows = list(cur.execute("select s.%s, d.start, d.end, d.grid_size_y from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s "
                            "on d.kernel_id = s.id order by d.start" % name_col))
    short = [re.sub(r"\(.*", "", n).replace("void ", "").replace("masp::", "") for n, *_ in rows]
    stages = sum(1 for s, r in zip(short, rows) if s.startswith("k_msm_combine<FpOps") and r[3] == 256)
    if not stages:
        sys.exit("no k_msm_combine<FpOps> dispatch with grid.y = 256: not a trace of full batches")
    # curve of the curve-independent kernels: that of the next k_tree_pass1 on the stream
    curve = [None] * len(rows)
    nxt = None
    for i in range(len(rows) - 1, -1, -1):
        if short[i].startswith("k_tree_pass1<"):
            nxt = "G1" if short[i].startswith("k_tree_pass1<FpOps") else "G2"
        curve[i] = nxt
    per_kernel, per_group = {}, {g: 0.0 for g, _ in GROUPS}
    for i, (s, r) in enumerate(zip(short, rows)):
        for g, names in GROUPS:       # (these kernels only run for batches of >= 8 proofs: a lone proof has no tree)
            for nm in names:
                if not s.startswith(nm):
                    continue
                if "<" in s:
                    if not s.startswith(nm + "<FpOps"):
                        continue
                elif curve[i] != "G1":
                    continue
                us = (r[2] - r[1]) / 1e3
                k = per_kernel.setdefault(s, [0, 0.0, g])
                k[0] += 1
                k[1] += us
                per_group[g] += us
    total = sum(per_group.values())
    print("# G1 bucket-accumulation stage per Spend MSM (one stage = one G1 MSM of a 256-proof batch), from %s" % path.split("/")[-1])
    print("# stages (k_msm_combine<FpOps>, grid.y = 256): %d" % stages)
    print("%-44s %-22s %8s %14s %12s" % ("kernel", "group", "calls", "total_us", "ms_per_MSM"))
    for s, (n, us, g) in sorted(per_kernel.items(), key=lambda kv: -kv[1][1]):
        print("%-44s %-22s %8d %14.1f %12.3f" % (s[:44], g, n, us, us / 1e3 / stages))
    print("#")
    print("%-44s %12s" % ("group (bench.py roofline.kernel_ms_per_launch)", "ms_per_MSM"))
    for g, _ in GROUPS:
        print("%-44s %12.3f" % (g, per_group[g] / 1e3 / stages))
    print("%-44s %12.3f   <- set against roofline.avg_launch_ms (HIP events around the same stage, which also span the" % ("SUM of kernel durations per stage", total / 1e3 / stages))
    print("%-44s %12s      gaps between ~70 launches and the stage's device-to-device copies)" % ("", ""))
    alg = 256 * 48725632 / 3.0
    print("# algorithmic bytes per stage (SURVEY.md 8d): 256 x 48 725 632 / 3 = %.0f B -> %.1f GB/s over the summed kernel time = %.5f of 8 000 GB/s"
          % (alg, alg / (total / 1e6 / stages) / 1e9, alg / (total / 1e6 / stages) / 1e9 / 8000.0))


if __name__ == "__main__":
    main(sys.argv[1])
