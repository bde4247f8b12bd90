#!/usr/bin/env python3
"""Per-API-call totals from a rocprofv3 --hip-runtime-trace rocpd sqlite file (count, total, average), restricted to
the last `tail` fraction of the trace.  usage: rocpd_api_summary.py results.db [tail_fraction]"""
import sqlite3
import sys


def main(path, tail=1.0):
    db = sqlite3.connect(path)
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
    reg = [t for t in tabs if t.startswith("rocpd_region")]
    if not reg:
        print("no region table; tables:", tabs); return
    cols = [r[1] for r in db.execute("pragma table_info(%s)" % reg[0])]
    print(reg[0], cols)
    strtab = [t for t in tabs if t.startswith("rocpd_string")][0]
    t0, t1 = db.execute("select min(start), max(end) from %s" % reg[0]).fetchone()
    cut = t1 - (t1 - t0) * tail
    q = ("select s.string, count(*), sum(r.end - r.start), avg(r.end - r.start), count(distinct r.tid) from %s r join %s s on r.name_id = s.id "
         "where r.start >= ? group by s.string order by 3 desc limit 25" % (reg[0], strtab))
    print("window %.1f ms" % ((t1 - cut) / 1e6))
    for name, n, tot, avg, nt in db.execute(q, (cut,)):
        print("%-40s calls %7d  total %9.2f ms  avg %8.2f us  threads %d" % (name[:40], n, tot / 1e6, avg / 1e3, nt))


if __name__ == "__main__":
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 1.0)
