#!/usr/bin/env python3
"""GPU occupancy in time from a rocprofv3 kernel trace (rocpd sqlite): over the window between two kernels'
first/last dispatches, how long at least one kernel was running (union of [start,end]) and the average number of
kernels in flight.  usage: rocpd_busy.py results.db [tail_fraction]   (tail_fraction: analyse only the last part
of the trace, e.g. 0.5, to skip set-up phases)"""
import sqlite3
import sys


def main(path, tail=1.0):
    db = sqlite3.connect(path)
    rows = db.execute("select start, end from rocpd_kernel_dispatch order by start").fetchall()
    if not rows:
        print("no dispatches"); return
    t0, t1 = rows[0][0], max(r[1] for r in rows)
    cut = t1 - (t1 - t0) * tail
    rows = [r for r in rows if r[0] >= cut]
    span = max(r[1] for r in rows) - rows[0][0]
    busy, cur_s, cur_e, total = 0, None, None, 0
    for s, e in rows:
        total += e - s
        if cur_e is None or s > cur_e:
            if cur_e is not None: busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    print("dispatches %d  span %.3f ms  busy(union) %.3f ms = %.1f %%  sum of durations %.3f ms  avg kernels in flight while busy %.2f"
          % (len(rows), span / 1e6, busy / 1e6, 100.0 * busy / span, total / 1e6, total / busy))


if __name__ == "__main__":
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 1.0)
