#!/bin/bash
# round 3, first GPU call: one-workgroup transforms vs two-pass -- equality, timing, full parity suite, counters, bench A/B
export TMPDIR=/tmp
R=$PWD; out=$R/gpurun_out/r03a; mkdir -p $out
timeout 300 $R/cuhe_amd/lib/ow_ab 4096 10 > $out/ow_ab.txt 2>&1; tail -20 $out/ow_ab.txt
timeout 900 python -m pytest tests -m gpu -q --maxfail=20 2>&1 | tail -60 > $out/pytest.txt; tail -5 $out/pytest.txt
cd /tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" \
           "GRBM_GUI_ACTIVE FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $set -d /tmp/pm_$i -o p -- $R/cuhe_amd/lib/ow_ab 512 2 > /dev/null 2>&1
  python $R/tools/rocpd_summary.py /tmp/pm_$i/p_results.db > $out/pmc_$i.txt 2>&1
done
cd $R
for m in 0 1; do
  CUHE_ONEWG=$m timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu --no-prince 2>/dev/null | tail -1 > $out/bench_onewg$m.json
done
python - <<PY
import json
for m in (0, 1):
    try:
        d = json.load(open("$out/bench_onewg%d.json" % m)); r = d["roofline"]
        print("onewg", m, "NTT/s", d["value"], "frac", r["frac"], "mul_relin ms", d["mul_relin"]["ms"], "batched", d["mul_relin"]["batched"]["ms_per_ciphertext"],
              "other ring", d["mul_relin_other_ring"]["batched"]["ms_per_ciphertext"], "mul_full ms", d["mul_full"]["ms"], "batched", d["mul_full"]["batched"]["ms_per_multiply"])
    except Exception as e:
        print("onewg", m, "bench failed", e)
PY
