#!/bin/bash
# End-of-round evidence run on the GPU box: test-suite, default bench line, kernel trace of the same command, the PMC
# passes of the metric's kernels (one counter family per pass, each with --kernel-trace only), and the kernel trace +
# HBM counters of the batched multiply + relinearise call.  Text summaries land in gpurun_out/final/;
# tools/make_traffic_json.py turns the PMC passes into profiles/traffic_rNN.json (bytes and VALU lane-instructions per
# transform, tagged with the hash of the kernel sources bench.py checks).  Every step runs under its own timeout.
# usage (from the repo root on the GPU box): tools/profile_final.sh [skip-tests] [round-tag, default r06]
export TMPDIR=/tmp
tag=${2:-r06}
out=$PWD/gpurun_out/final; mkdir -p $out
if [ "$1" != "skip-tests" ]; then timeout 1500 python -m pytest tests -m gpu -q --durations=8 --timeout 400 2>&1 | tail -40 > $out/pytest_gpu.txt; cat $out/pytest_gpu.txt; fi
timeout 600 python bench.py 2>/dev/null | tail -1 > $out/bench_n1.json
python - <<PY
import json; d = json.load(open("$out/bench_n1.json")); r = d["roofline"]
print("NTT/s", d["value"], "frac", r["frac"], "other lengths", {k: (v.get("value"), v.get("frac")) for k, v in r.get("other_lengths", {}).items()}, "copy GB/s", r.get("measured_copy_GBs"))
print("mul_relin (x^65536+1) ms", d["mul_relin"]["ms"], "batched", d["mul_relin"]["batched"]["ms_per_ciphertext"], d["mul_relin"]["batched"].get("checked"),
      "| x^32768+1", d["mul_relin_other_ring"]["ms"], d["mul_relin_other_ring"]["batched"]["ms_per_ciphertext"], "| mul_full ms", d["mul_full"]["ms"], "batched", d["mul_full"]["batched"]["ms_per_multiply"], "| prince", (d.get("prince") or {}).get("value"))
PY
R=$PWD
cd /tmp
rm -rf /tmp/pf_*
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pf_stats -o s -- python $R/bench.py --steps 10 --warmup 2 --no-cpu --no-prince --no-limiter > /dev/null 2>&1
python $R/tools/rocpd_summary.py /tmp/pf_stats/s_results.db > $out/kernel_trace_stats.txt 2>&1
for c in FETCH_SIZE WRITE_SIZE SQ_INSTS_VALU; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c -d /tmp/pf_$c -o p -- python $R/bench.py --steps 2 --warmup 1 --no-mulrelin --no-cpu --no-prince --no-limiter > /dev/null 2>&1
  python $R/tools/rocpd_summary.py /tmp/pf_$c/p_results.db 2>&1 | grep -E "^==|^kernel|ntt_pass|ntt_onewg" > $out/pmc_$c.txt
done
# the batched multiply + relinearise call alone (config 4, 32 ciphertexts per call): per-kernel time, then the HBM
# counters of its inner-product kernel
timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/pf_batched -o s -- python $R/tools/trace_batched.py 32 10 2>&1 | grep -v "^[WEI][0-9]\{8\} " > $out/batched_trace.txt
python $R/tools/rocpd_summary.py /tmp/pf_batched/s_results.db 2>&1 | head -16 | cut -c1-84,112-200 >> $out/batched_trace.txt
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --kernel-trace --pmc $c -d /tmp/pf_b$c -o p -- python $R/tools/trace_batched.py 32 3 > /dev/null 2>&1
  python $R/tools/rocpd_summary.py /tmp/pf_b$c/p_results.db 2>&1 | grep -E "$c" | grep -E "k_relin_mac_mfma|k_icrt|ntt_pass1w<15, 4>" | cut -c1-60,100-200 >> $out/batched_trace.txt
done
cd $R
python tools/make_traffic_json.py $out $tag > $out/traffic_$tag.json && cat $out/traffic_$tag.json
# the one-workgroup transforms against the two-pass kernels (equality + timing), and the table of doc/Perf_NTT.txt
[ -x $R/cuhe_amd/lib/ow_ab ] || hipcc -O2 $R/tools/ow_ab.cpp -I$R/include -L$R/cuhe_amd/lib -lcuhe_hip -Wl,-rpath,$R/cuhe_amd/lib -o $R/cuhe_amd/lib/ow_ab
timeout 300 $R/cuhe_amd/lib/ow_ab 4096 10 > $out/onewg_ab.txt 2>&1; grep -v mismatch $out/onewg_ab.txt; grep -c identical $out/onewg_ab.txt
timeout 300 python bench.py --perf-table $out/perf_ntt_table.txt > /dev/null 2>&1; tail -11 $out/perf_ntt_table.txt
head -14 $out/kernel_trace_stats.txt | cut -c1-72,110-200
cat $out/batched_trace.txt
