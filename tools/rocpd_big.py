#!/usr/bin/env python3
"""Per-kernel totals of a rocprofv3 kernel trace (rocpd sqlite) restricted to dispatches of at least `min_threads`
work-items: separates the large array launches of a batched run from the small per-ciphertext ones around them.
usage: rocpd_big.py results.db [min_threads]"""
import sqlite3
import sys


def main(path, min_threads=200000):
    db = sqlite3.connect(path)
    rows = db.execute(
        "select s.display_name, count(*), sum(d.end-d.start), avg(d.end-d.start) from rocpd_kernel_dispatch d "
        "join rocpd_info_kernel_symbol s on d.kernel_id = s.id where d.grid_size_x * d.grid_size_y * d.grid_size_z >= ? "
        "group by 1 order by 3 desc", (min_threads,)).fetchall()
    tot = sum(r[2] for r in rows) or 1
    print("dispatches with >= %d work-items: total %.3f ms" % (min_threads, tot / 1e6))
    for r in rows[:16]:
        print("%-74s calls %5d  total %8.3f ms  avg %8.1f us  %5.1f %%" % (r[0][:74], r[1], r[2] / 1e6, r[3] / 1e3, 100.0 * r[2] / tot))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 200000)
