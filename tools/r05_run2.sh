#!/bin/bash
# round 5, GPU run 2: where the scheduled block spends its host time (cold vs warm, sampling profile, HIP runtime trace), the fused
# single-ciphertext chain (tests + bench), the new / changed GPU tests
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
L=cuhe_amd/lib
export TMPDIR=/tmp
R=$PWD
( timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_config4.py -m gpu -x -q -k "fused or relin_vs_oracle or dense or list_block or rccl" 2>&1 | tail -8 ) > gpurun_out/r05_pytest_new.txt
{
for cfg in "1 3" "1 2" "0 3" "2 3"; do
  set -- $cfg
  echo "== policy $1, $2 workers, 4 blocks in one process (cold, then warm)"
  CUHE_SCHED_STATS=1 CUHE_SCHED_POLICY=$1 timeout 300 $L/test_prince_flow --threads 1 --sched $2 --no-round-checks --repeat 4 2>&1 | grep -E "Prince Encryption|batches:|homomorphic PRINCE"
done
echo "== policy 1, 3 workers: sampling profile of the cold block"
CUHE_SCHED_POLICY=1 timeout 300 $L/test_prince_flow --threads 1 --sched 3 --no-round-checks --profile 2>&1 | python tools/resolve_samples.py | head -120
} > gpurun_out/r05_sched_cold_warm.txt 2>&1
( cd /tmp && rm -rf /tmp/ph && CUHE_SCHED_POLICY=1 timeout 300 rocprofv3 --hip-runtime-trace --stats -d /tmp/ph -o h -- $R/$L/test_prince_flow --threads 1 --sched 3 --no-round-checks 2>&1 | grep -E "Prince Encryption|PASSED"; ls /tmp/ph; python - <<'PY'
import sqlite3, glob
for f in glob.glob("/tmp/ph/*.db"):
    db = sqlite3.connect(f)
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
    reg = [t for t in tabs if t.startswith("rocpd_region") and "ext" not in t]
    strs = [t for t in tabs if t.startswith("rocpd_string")]
    print(f, reg[:2], strs[:1])
    try:
        q = "select s.string, count(*), sum(r.end - r.start), max(r.end - r.start) from %s r join %s s on r.name_id = s.id group by s.string order by 3 desc limit 25" % (reg[0], strs[0])
        for row in db.execute(q): print("%-40s %7d calls %10.3f ms total %9.3f ms max" % (row[0][:40], row[1], row[2] / 1e6, row[3] / 1e6))
    except Exception as ex:
        print("query failed:", ex, tabs[:30])
PY
) > gpurun_out/r05_sched_hiptrace.txt 2>&1
( timeout 600 python bench.py --no-prince 2>&1 | tail -2 ) > gpurun_out/r05_bench2.txt
cat gpurun_out/r05_pytest_new.txt; cat gpurun_out/r05_sched_cold_warm.txt | head -140; cat gpurun_out/r05_sched_hiptrace.txt | head -40
python - <<'PY'
import json
l=[x for x in open("gpurun_out/r05_bench2.txt") if x.startswith("{")]
if l:
    d=json.loads(l[-1])
    for k in ("mul_relin","mul_relin_other_ring"):
        m=d.get(k) or {}
        print(k, m.get("ms"), m.get("variants"), (m.get("batched") or {}).get("ms_per_ciphertext"))
    print("value", d["value"], d["roofline"]["frac"])
else:
    print(open("gpurun_out/r05_bench2.txt").read()[-2000:])
PY
