export TMPDIR=/tmp
R=$PWD
make -C cuhe_amd/cxx -s test > /dev/null 2>&1
cd /tmp; rm -rf /tmp/pp
timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/pp -o s -- $R/cuhe_amd/lib/test_prince_flow --threads 1 --no-round-checks 2>&1 | grep -E "Prince Encryption|PASSED"
python $R/tools/rocpd_summary.py /tmp/pp/s_results.db 2>&1 | head -24 | cut -c1-70,112-175
python - <<PY
import sqlite3, glob
db = sqlite3.connect("/tmp/pp/s_results.db")
t = [r[0] for r in db.execute("select name from sqlite_master where type='table' and name like 'rocpd_kernel_dispatch%'")]
n, tot = db.execute("select count(*), sum(end-start) from %s" % t[0]).fetchone()
print("kernel launches", n, "total kernel time s", tot / 1e9)
PY
