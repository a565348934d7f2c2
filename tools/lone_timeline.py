#!/usr/bin/env python3
"""Timeline of the last lone-proof run in a rocprofv3 rocpd database: every dispatch between the last single-block
k_groth16_finish_ac and `window_ms` before its end, with start / duration relative to the first of them and its queue.
usage: lone_timeline.py results.db [window_ms=40]"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
win = float(sys.argv[2]) if len(sys.argv) > 2 else 40.0
cur = db.cursor()
sym_cols = [r[1] for r in cur.execute("pragma table_info(rocpd_info_kernel_symbol)")]
name_col = "display_name" if "display_name" in sym_cols else "kernel_name"
rows = list(cur.execute("select s.%s, d.start, d.end, d.queue_id, d.grid_size_x, d.grid_size_y from rocpd_kernel_dispatch d "
                        "join rocpd_info_kernel_symbol s on d.kernel_id = s.id order by d.start" % name_col))
last = [r for r in rows if "k_groth16_finish_ac" in r[0] and r[4] <= 128][-1]   # grid of one workgroup: a lone proof
t1 = last[2]
sel = [r for r in rows if r[1] >= t1 - win * 1e6 and r[2] <= t1]
# keep only the run that belongs to this proof: starts at the last k_fr_to_mont / k_fr_split_forms before the assemble
starts = [r for r in sel if "k_fr_to_mont" in r[0] or "k_fr_split_forms" in r[0]]
t0 = starts[-1][1] if starts else sel[0][1]
print("%-58s %9s %9s %6s %s" % ("kernel", "start_us", "dur_us", "queue", "grid"))
for name, st, en, q, gx, gy in sel:
    if st < t0:
        continue
    short = re.sub(r"\(.*", "", name).replace("void masp::", "").replace("masp::", "")[:58]
    print("%-58s %9.1f %9.1f %6s %dx%d" % (short, (st - t0) / 1e3, (en - st) / 1e3, q, gx, gy))
print("# total %.2f ms" % ((t1 - t0) / 1e6))
