#!/usr/bin/env python3
"""Concurrency statistics out of a rocprofv3 rocpd database of the bench: over the middle of the longest burst of batched
dispatches (the timed region), what share of the time is at least one bucket accumulation resident, how many run at once,
and what runs when none does.  usage: timeline_stats.py results.db [grid_y=128]"""
import sqlite3
import sys


def main(path, gy):
    db = sqlite3.connect(path)
    cur = db.cursor()
    sym_cols = [r[1] for r in cur.execute("pragma table_info(rocpd_info_kernel_symbol)")]
    name_col = "display_name" if "display_name" in sym_cols else "kernel_name"
    rows = list(cur.execute("select s.%s, d.start, d.end, d.grid_size_y, d.queue_id from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s "
                            "on d.kernel_id = s.id order by d.start" % name_col))
    big = [r for r in rows if r[3] == gy]
    # bursts = maximal runs of batched dispatches without a gap > 20 ms
    bursts, cur_b = [], [big[0]]
    for r in big[1:]:
        if r[1] - max(x[2] for x in cur_b[-50:]) > 20e6:
            bursts.append(cur_b)
            cur_b = []
        cur_b.append(r)
    bursts.append(cur_b)
    b = max(bursts, key=len)
    t0, t1 = b[0][1], max(x[2] for x in b)
    lo, hi = t0 + 0.2 * (t1 - t0), t0 + 0.8 * (t1 - t0)
    print("longest burst: %d dispatches over %.1f ms; statistics over its middle 60 %% (%.1f ms)" % (len(b), (t1 - t0) / 1e6, (hi - lo) / 1e6))

    def short(n):
        n = n.split("(")[0].replace("void masp::", "").replace("masp::", "")
        return n
    ev = []
    for n, s, e, _, q in b:
        s, e = max(s, lo), min(e, hi)
        if e > s:
            ev.append((s, 1, short(n)))
            ev.append((e, -1, short(n)))
    ev.sort(key=lambda x: (x[0], x[1]))
    live = {}
    last = lo
    acc_hist, none_what, any_busy = {}, {}, 0
    for t, d, n in ev:
        dt = t - last
        if dt > 0:
            nacc = sum(v for k, v in live.items() if "k_msm_accumulate" in k)
            acc_hist[nacc] = acc_hist.get(nacc, 0) + dt
            if sum(live.values()) > 0:
                any_busy += dt
            if nacc == 0:
                key = "+".join(sorted(k for k, v in live.items() if v > 0)) or "(idle)"
                none_what[key] = none_what.get(key, 0) + dt
        live[n] = live.get(n, 0) + d
        last = t
    tot = hi - lo
    print("accumulations resident at once: " + ", ".join("%d: %.1f %%" % (k, 100.0 * v / tot) for k, v in sorted(acc_hist.items())))
    print("some kernel resident: %.1f %% of the time" % (100.0 * any_busy / tot))
    print("while NO accumulation is resident (%.1f %%), the device runs:" % (100.0 * acc_hist.get(0, 0) / tot))
    for k, v in sorted(none_what.items(), key=lambda x: -x[1])[:14]:
        print("  %5.1f %%  %s" % (100.0 * v / tot, k[:150]))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 128)
