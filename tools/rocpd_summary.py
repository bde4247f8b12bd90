#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd (sqlite) result: per-kernel call count / total / average
duration (the `--kernel-trace --stats` table) and, if present, per-dispatch PMC averages.
usage: rocpd_summary.py results.db [more.db ...]"""
import sqlite3
import sys


def short(name):
    name = name.replace("(anonymous namespace)::", "")
    return name if len(name) < 110 else name[:107] + "..."


def main(paths):
    for p in paths:
        db = sqlite3.connect(p)
        print("== %s" % p)
        rows = db.execute(
            "select s.display_name, count(*), sum(d.end-d.start), avg(d.end-d.start), min(d.end-d.start), max(d.end-d.start),"
            " max(s.arch_vgpr_count), max(s.sgpr_count), max(d.group_segment_size), max(d.grid_size_x), max(d.workgroup_size_x)"
            " from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id"
            " group by s.display_name order by 3 desc").fetchall()
        tot = sum(r[2] for r in rows) or 1
        print("%-110s %6s %12s %10s %10s %10s %6s %5s %5s %7s" % ("kernel", "calls", "total_ns", "avg_ns", "min_ns", "max_ns", "pct", "vgpr", "sgpr", "lds"))
        for r in rows:
            print("%-110s %6d %12d %10.0f %10d %10d %6.2f %5s %5s %7s" % (short(r[0]), r[1], r[2], r[3], r[4], r[5], 100.0 * r[2] / tot, r[6], r[7], r[8]))
        try:
            pm = db.execute(
                "select s.display_name, i.name, count(*), avg(e.value), sum(e.value) from rocpd_pmc_event e"
                " join rocpd_info_pmc i on e.pmc_id = i.id"
                " join rocpd_kernel_dispatch d on d.event_id = e.event_id"
                " join rocpd_info_kernel_symbol s on d.kernel_id = s.id group by s.display_name, i.name order by 5 desc").fetchall()
        except sqlite3.Error as ex:
            pm = []
            print("(no pmc table: %s)" % ex)
        if pm:
            print("-- PMC (per dispatch average)")
            for r in pm:
                print("%-110s %-14s n=%-5d avg=%14.1f sum=%16.1f" % (short(r[0]), r[1], r[2], r[3], r[4]))


if __name__ == "__main__":
    main(sys.argv[1:])
