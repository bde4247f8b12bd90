#!/bin/bash
# round 5, GPU run 3: scheduled PRINCE after the retire fix (cold + warm blocks), workers x batch-workers, then tests of the changed entry points
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
L=cuhe_amd/lib
export TMPDIR=/tmp
R=$PWD
{
for cfg in "2 0" "3 0" "3 1" "2 1" "4 1" "3 2"; do
  set -- $cfg
  for rep in 1 2; do
  echo "== policy 1, $1 workers, batch workers $2: sync block, then 4 scheduled blocks"
  CUHE_SCHED_STATS=1 CUHE_SCHED_BATCH_WORKERS=$2 timeout 300 $L/test_prince_flow --threads 1 --sched $1 --no-round-checks --repeat 4 2>&1 | grep -E "Prince Encryption|batches:|scheduler:" | cut -c1-200
  done
done
echo "== policy 0 (round 4), 3 workers"
CUHE_SCHED_STATS=1 CUHE_SCHED_POLICY=0 timeout 300 $L/test_prince_flow --threads 1 --sched 3 --no-round-checks --repeat 4 2>&1 | grep -E "Prince Encryption|batches:" | cut -c1-200
echo "== compare (what bench.py runs)"
timeout 300 $L/test_prince_flow --threads 1 --no-round-checks --compare --repeat 4 2>&1 | grep -E "Prince Encryption|homomorphic"
echo "== 8 / 3 virtual devices, 1 device, policy 1, 4 blocks"
for d in 8 3; do CUHE_SCHED_STATS=1 timeout 300 $L/test_prince_flow --threads 1 --sched 3 --no-round-checks --devices $d --virtual --repeat 3 2>&1 | grep -E "Prince Encryption|batches:|homomorphic PRINCE" | cut -c1-200; done
} > gpurun_out/r05_sched_run3.txt 2>&1
( cd /tmp && rm -rf /tmp/ph && timeout 300 rocprofv3 --hip-runtime-trace --stats -d /tmp/ph -o h -- $R/$L/test_prince_flow --threads 1 --sched 3 --no-round-checks 2>&1 | grep -E "Prince Encryption|PASSED"; python - <<'PY'
import sqlite3, glob
for f in glob.glob("/tmp/ph/*.db"):
    db = sqlite3.connect(f)
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
    reg = [t for t in tabs if t.startswith("rocpd_region") and "ext" not in t]
    strs = [t for t in tabs if t.startswith("rocpd_string")]
    q = "select s.string, count(*), sum(r.end - r.start), max(r.end - r.start) from %s r join %s s on r.name_id = s.id group by s.string order by 3 desc limit 12" % (reg[0], strs[0])
    for row in db.execute(q): print("%-40s %7d calls %10.3f ms total %9.3f ms max" % (row[0][:40], row[1], row[2] / 1e6, row[3] / 1e6))
PY
) > gpurun_out/r05_sched_hiptrace2.txt 2>&1
( timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_config4.py tests/test_gpu_cxx_api.py -m gpu -x -q -k "fused or relin_vs_oracle or dense or scheduled" 2>&1 | tail -6 ) > gpurun_out/r05_pytest_new2.txt
cat gpurun_out/r05_sched_run3.txt; cat gpurun_out/r05_sched_hiptrace2.txt; cat gpurun_out/r05_pytest_new2.txt
