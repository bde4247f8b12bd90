#!/bin/bash
# One GPU-box pass, by name (replaces the per-run scratch scripts of earlier rounds).  usage: tools/gpu_pass.sh <what> [args]
#   suite            the whole -m gpu suite with the 30 slowest durations            -> gpurun_out/pytest_gpu.txt
#   bench            the default bench line + a digest of it                         -> gpurun_out/bench_n1.json
#   allocfail        allocation-failure injection at ten points, scheduled + sync    -> gpurun_out/allocfail.txt
#   probes           copy-ceiling shapes, host thread scaling, mulZZX staging        -> gpurun_out/probes.txt   (tools/gpu_probes.py)
#   soak [reps]      tools/sched_soak.sh
#   rebase-ab        tools/rebase_arg_ab.py (needs cuhe_amd/lib/libcuhe_hip_norebase.so from tools/build_variant.py)
#   final [tag]      tools/profile_final.sh skip-tests <tag>: kernel trace, PMC passes, traffic json, perf table
mkdir -p gpurun_out; export TMPDIR=/tmp
L=cuhe_amd/lib
case "$1" in
suite)
  timeout 1700 python -m pytest tests -q -m gpu --durations=30 > gpurun_out/pytest_gpu.txt 2>&1; tail -45 gpurun_out/pytest_gpu.txt ;;
bench)
  ( timeout 1500 python bench.py 2> gpurun_out/bench_stderr.txt | tail -1 ) > gpurun_out/bench_n1.json
  python - <<'PY'
import json
d=json.loads(open("gpurun_out/bench_n1.json").read().strip().splitlines()[-1]); r=d["roofline"]
print("value", d["value"], "frac", r["frac"], "frac_from_ms_per_step", r.get("frac_from_ms_per_step"), "ms_per_step", d["ms_per_step"], "traffic", r["traffic"])
print("copy", r.get("measured_copy_GBs"), r.get("measured_copy_variants_GBs"))
print("cpu", json.dumps(d["cpu_baseline"]))
for k in ("mul_relin","mul_relin_other_ring"):
    m=d[k]
    if m: print(k, m.get("ms"), m.get("cpu_baseline"), (m.get("batched") or {}).get("ms_per_ciphertext"), ((m.get("batched") or {}).get("twice_the_batch") or {}).get("ms_per_ciphertext"))
print("mul_full", d["mul_full"]["ms"], d["mul_full"]["batched"]["ms_per_multiply"], d["mul_full"].get("cpu_baseline"))
print("prince", d["prince"]["value"], d["prince"].get("later_blocks_same_process"), json.dumps(d["prince"]["gate_by_gate"])[:1500])
PY
  tail -3 gpurun_out/bench_stderr.txt ;;
allocfail)
  OUT=gpurun_out/allocfail.txt; : > $OUT
  make -C cuhe_amd/cxx -s test >> $OUT 2>&1
  for N in 0 1 2 3 7 20 40 80 150 200; do for S in unset 0; do
    if [ $S = 0 ]; then export CUHE_SCHED=0; else unset CUHE_SCHED; fi
    s=$(date +%s.%N); timeout 120 $L/test_sched_soak allocfail $N > gpurun_out/soak_one.log 2>&1; rc=$?
    echo "allocation failure injected at allocation $N, CUHE_SCHED $S: exit $rc ($(grep -c 'cuheSafeCall() failed' gpurun_out/soak_one.log) cuheSafeCall message(s)), $(python3 -c "import time,sys; print('%.1f' % (time.time()-float(sys.argv[1])))" $s) s" >> $OUT
  done; done
  unset CUHE_SCHED; cat $OUT ;;
probes) python tools/gpu_probes.py > gpurun_out/probes.txt 2>&1; cat gpurun_out/probes.txt ;;
soak) tools/sched_soak.sh ${2:-20} ;;
rebase-ab) python tools/rebase_arg_ab.py > gpurun_out/rebase_arg_ab.txt 2>&1; cat gpurun_out/rebase_arg_ab.txt ;;
final) tools/profile_final.sh skip-tests ${2:-r06} ;;
*) sed -n 2,11p "$0" ;;
esac
