#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd (sqlite) result: per-kernel call count / total / average
duration (the `--kernel-trace --stats` table) and, if present, per-dispatch PMC averages.
usage: rocpd_summary.py results.db [more.db ...]
       rocpd_summary.py --between MARKER results.db     per-kernel table of every stretch of dispatches enclosed by two runs of the
                                                        kernel MARKER (the clients' CUHE_TRACE_MARK=1: k_probe_valu around the timed part)"""
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


def between(marker, path, top=40):
    db = sqlite3.connect(path)
    rows = db.execute("select s.display_name, d.start, d.end from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s"
                      " on d.kernel_id = s.id order by d.start").fetchall()
    windows, cur, seen_marker = [], None, False
    for name, st, en in rows:
        if marker in name:
            if cur:
                windows.append(cur)
            cur, seen_marker = None, True
        elif seen_marker:
            cur = cur or []
            cur.append((name, st, en))
    print("== %s: %d stretches enclosed by '%s'" % (path, len(windows), marker))     # (what follows the last marker is not enclosed)
    for wi, w in enumerate(windows):
        span = max(e for _, _, e in w) - min(s for _, s, _ in w)
        # time with at least one kernel running (union of the intervals) and the sum of the durations (> union where streams overlap)
        busy, last = 0, 0
        for _, st, en in sorted(w, key=lambda x: x[1]):
            if en > last:
                busy += en - max(st, last)
                last = en
        agg = {}
        for name, st, en in w:
            a = agg.setdefault(short(name), [0, 0])
            a[0] += 1
            a[1] += en - st
        tot = sum(a[1] for a in agg.values())
        print("-- stretch %d: %d dispatches, first start to last end %.3f ms, some kernel running %.3f ms, sum of durations %.3f ms"
              % (wi, len(w), span / 1e6, busy / 1e6, tot / 1e6))
        for name, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
            print("%-110s %6d %12d %10.0f %6.2f" % (name, a[0], a[1], a[1] / a[0], 100.0 * a[1] / tot))
        # where the device sat idle: gaps between the end of everything running and the next start, by size and by the kernel that follows
        gaps, last, prev = [], None, None
        for name, st, en in sorted(w, key=lambda x: x[1]):
            if last is not None and st > last:
                gaps.append((st - last, short(prev)[:48], short(name)[:48]))
            if last is None or en > last:
                last, prev = en, name
        if gaps:
            edges = [2e3, 5e3, 10e3, 20e3, 50e3, 100e3, 1e9]
            hist = [[0, 0] for _ in edges]
            for g in gaps:
                for i, e in enumerate(edges):
                    if g[0] <= e:
                        hist[i][0] += 1
                        hist[i][1] += g[0]
                        break
            print("   idle gaps: %d, %.3f ms in total; by size (us): %s" % (len(gaps), sum(g[0] for g in gaps) / 1e6, "  ".join(
                "<=%g: %d (%.2f ms)" % (e / 1e3, h[0], h[1] / 1e6) for e, h in zip(edges, hist) if h[0])))
            by_next = {}
            for g in gaps:
                a = by_next.setdefault(g[2], [0, 0])
                a[0] += 1
                a[1] += g[0]
            for name, a in sorted(by_next.items(), key=lambda kv: -kv[1][1])[:12]:
                print("   idle before %-50s %5d gaps %9.3f ms  (%.1f us each)" % (name, a[0], a[1] / 1e6, a[1] / a[0] / 1e3))
            for g in sorted(gaps, reverse=True)[:8]:
                print("   gap %8.1f us  after %-48s before %s" % (g[0] / 1e3, g[1], g[2]))


if __name__ == "__main__":
    if len(sys.argv) > 3 and sys.argv[1] == "--between":
        for p in sys.argv[3:]:
            between(sys.argv[2], p)
    else:
        main(sys.argv[1:])
