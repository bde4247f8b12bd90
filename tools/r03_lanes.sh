#!/bin/bash
export TMPDIR=/tmp
R=$PWD; out=$R/gpurun_out/r03g; mkdir -p $out
for l in 1 2 3; do
  CUHE_RELIN_LANES=$l timeout 400 python bench.py --steps 5 --warmup 2 --no-cpu --no-prince 2>/dev/null | tail -1 > $out/bench_lanes$l.json
  CUHE_RELIN_LANES=$l CUHE_ONEWG=0 timeout 400 python bench.py --steps 5 --warmup 2 --no-cpu --no-prince 2>/dev/null | tail -1 > $out/bench_lanes${l}_onewg0.json
done
python - <<PY
import json
for n in ("lanes1", "lanes1_onewg0", "lanes2", "lanes2_onewg0", "lanes3", "lanes3_onewg0"):
    try:
        d = json.load(open("$out/bench_%s.json" % n))
        print(n, "mul_relin batched", d["mul_relin"]["batched"]["ms_per_ciphertext"], d["mul_relin"]["batched"].get("checked"), "other ring", d["mul_relin_other_ring"]["batched"]["ms_per_ciphertext"])
    except Exception as e:
        print(n, "failed", e)
PY
