#!/bin/bash
export TMPDIR=/tmp
R=$PWD; out=$R/gpurun_out/r03d; mkdir -p $out
timeout 1200 python -m pytest tests -m gpu -q --maxfail=10 2>&1 | tail -40 > $out/pytest.txt; grep -E "passed|failed" $out/pytest.txt
for m in "0 0" "1 0"; do
  set -- $m
  CUHE_ONEWG=$1 CUHE_ONEWG64=$2 timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu --no-prince 2>/dev/null | tail -1 > $out/bench_onewg$1$2.json
done
python - <<PY
import json
for m in ("00", "10"):
    try:
        d = json.load(open("$out/bench_onewg%s.json" % m)); r = d["roofline"]
        print("onewg", m, "NTT/s", d["value"], "frac", r["frac"], "mul_relin ms", d["mul_relin"]["ms"], "batched", d["mul_relin"]["batched"]["ms_per_ciphertext"],
              "other ring", d["mul_relin_other_ring"]["ms"], d["mul_relin_other_ring"]["batched"]["ms_per_ciphertext"], "mul_full ms", d["mul_full"]["ms"], "batched", d["mul_full"]["batched"]["ms_per_multiply"])
    except Exception as e:
        print("onewg", m, "bench failed", e)
PY
