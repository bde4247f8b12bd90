#!/bin/bash
# round 5, GPU run 6: persistent split inverse of 64K-point rows (CUHE_ONEWG_SPLIT=3) -- parity, then A/B on the batched multiply + relinearise of
# x^65536+1; then the bench line with the PMC passes measured live
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
{
echo "== parity with CUHE_ONEWG_SPLIT=3"
CUHE_ONEWG_SPLIT=3 timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_config4.py -m gpu -x -q -k "degree_65536 or (dense and 65536) or mul_relin_batch_equals_single or config4_relin or matrix_core" 2>&1 | tail -5
for rep in 1 2 3; do for m in 1 3; do
  echo "== CUHE_ONEWG_SPLIT=$m, batch 32, x^65536+1"
  CUHE_ONEWG_SPLIT=$m timeout 200 python tools/trace_batched.py 32 10 2^16 2>&1 | grep -v "^[WEI][0-9]\{8\} " | tail -2
done; done
} > gpurun_out/r05_split_inv_ab.txt 2>&1
cat gpurun_out/r05_split_inv_ab.txt
R=$PWD
( cd /tmp && for m in 1 3; do rm -rf /tmp/pb$m; CUHE_ONEWG_SPLIT=$m timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/pb$m -o s -- python $R/tools/trace_batched.py 32 10 2^16 > /dev/null 2>&1; echo "== split mode $m"; python $R/tools/rocpd_summary.py /tmp/pb$m/s_results.db 2>&1 | head -9 | cut -c1-84,112-200; done ) > gpurun_out/r05_split_inv_trace.txt 2>&1
cat gpurun_out/r05_split_inv_trace.txt
( timeout 900 python bench.py 2>&1 | tail -1 ) > gpurun_out/r05_bench_live.json
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r05_bench_live.json").read().strip().splitlines()[-1])
r=d["roofline"]
print("value", d["value"], "frac", r["frac"], "traffic", r["traffic"]); print(r["traffic_note"][:400]); print(r.get("traffic_live")); print(r["valu_ceiling"].get("achieved_T_per_s"), r["valu_ceiling"].get("frac_of_live_dense_stream"))
print("prince", d["prince"]["value"], d["prince"]["gate_by_gate"])
PY
