#!/bin/bash
# round 5, last GPU run: the GPU suite and the default bench line on the final tree
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out/final2
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests -m gpu -q --durations=6 --timeout 400 2>&1 | tail -14 ) > gpurun_out/final2/pytest_gpu.txt
( timeout 900 python bench.py 2>/dev/null | tail -1 ) > gpurun_out/final2/bench_n1.json
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/final2/smoke.txt 2>&1
cat gpurun_out/final2/pytest_gpu.txt gpurun_out/final2/smoke.txt
python - <<'PY'
import json
d=json.loads(open("gpurun_out/final2/bench_n1.json").read().strip().splitlines()[-1]); r=d["roofline"]
print("value", d["value"], "frac", r["frac"], "ms_per_step", d["ms_per_step"], "traffic", r["traffic"], r["traffic_note"][:60])
print("others", {k:(v.get("value"), v.get("frac")) for k,v in r["other_lengths"].items()})
for k in ("mul_relin","mul_relin_other_ring"):
    m=d[k]; print(k, m["ms"], m.get("variants"), m["batched"]["ms_per_ciphertext"], (m["batched"].get("twice_the_batch") or {}).get("ms_per_ciphertext"), (m.get("concurrent") or {}).get("ms_per_ciphertext"))
print("mul_full", d["mul_full"]["ms"], d["mul_full"]["batched"]["ms_per_multiply"])
print("prince", d["prince"]["value"], d["prince"]["gate_by_gate"])
print("cpu", d["cpu_baseline"]); print("live", json.dumps(r["valu_ceiling"]["live"])[:900])
PY
