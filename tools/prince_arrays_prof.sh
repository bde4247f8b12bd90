#!/bin/bash
# kernel trace of the homomorphic PRINCE on arrays of ciphertexts (bench.py's prince leg)
export TMPDIR=/tmp
R=$PWD
cd /tmp; rm -rf /tmp/pa
timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/pa -o s -- $R/cuhe_amd/lib/test_prince_arrays_cxx --no-round-checks --async --json --devices 1 2>&1 | grep -E "^\{" | cut -c1-200
python $R/tools/rocpd_summary.py /tmp/pa/s_results.db 2>&1 | head -26 | cut -c1-70,112-175
python - <<PY
import sqlite3
db = sqlite3.connect("/tmp/pa/s_results.db")
t = [r[0] for r in db.execute("select name from sqlite_master where type='table' and name like 'rocpd_kernel_dispatch%'")]
n, tot, t0, t1 = db.execute("select count(*), sum(end-start), min(start), max(end) from %s" % t[0]).fetchone()
print("kernel launches", n, "total kernel time s", tot / 1e9, "span s", (t1 - t0) / 1e9)
PY
