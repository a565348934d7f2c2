#!/usr/bin/env python3
"""Kernel statistics (the `--stats` view) out of a rocprofv3 rocpd SQLite database.
usage: rocpd_stats.py results.db [grid_y] [> profiles/summary.txt]
grid_y: only dispatches with that gridDim.y (= proofs per launch), which isolates the full batches of the timed region."""
import re
import sqlite3
import sys


def main(path, gy=None):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(rocpd_kernel_dispatch)")]
    sym_cols = [r[1] for r in cur.execute("pragma table_info(rocpd_info_kernel_symbol)")]
    name_col = "display_name" if "display_name" in sym_cols else "kernel_name"
    q = ("select s.%s, count(*), sum(d.end - d.start), avg(d.end - d.start), min(d.end - d.start), max(d.end - d.start) "
         "from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id %s group by s.%s order by 3 desc"
         % (name_col, ("where d.grid_size_y = %d" % gy) if gy else "", name_col))
    rows = list(cur.execute(q))
    total = sum(r[2] for r in rows) or 1
    print("%-70s %8s %12s %12s %10s %10s %6s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "%"))
    for name, n, tot, avg, mn, mx in rows:
        short = re.sub(r"\(.*", "", name)[:70]
        print("%-70s %8d %12.1f %12.2f %10.2f %10.2f %6.2f" % (short, n, tot / 1e3, avg / 1e3, mn / 1e3, mx / 1e3, 100.0 * tot / total))
    # the dominant kernel by batch size (grid.y = proofs per launch): bench.py's live roofline leg times only the launches
    # of its timed region, i.e. the rows with the full batch size
    q2 = ("select d.grid_size_y, count(*), avg(d.end - d.start) from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s "
          "on d.kernel_id = s.id where s.%s like '%%k_msm_accumulate<masp::FpOps>%%' group by d.grid_size_y order by 1" % name_col)
    print("# k_msm_accumulate<G1> by proofs per launch (grid.y): " +
          ", ".join("P=%d: %d launches, avg %.2f us" % (gy, n, avg / 1e3) for gy, n, avg in cur.execute(q2)))
    t0, t1 = cur.execute("select min(start), max(end) from rocpd_kernel_dispatch").fetchone()
    print("# span of all dispatches: %.3f ms; summed kernel time: %.3f ms" % ((t1 - t0) / 1e6, total / 1e6))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else None)
