#!/bin/bash
# after the one-workgroup tables moved to init: PRINCE on arrays (first block of a process), scheduled gate by gate (4 blocks), the gaps again,
# and the worker sweep of the scheduler
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
R=$PWD; L=cuhe_amd/lib
for i in 1 2 3; do timeout 300 $L/test_prince_arrays_cxx --no-round-checks --async 2>&1 | grep -E "Prince Encryption|FAILED"; done
timeout 300 $L/test_prince_arrays_cxx 2>&1 | grep -E "Prince Encryption|FAILED|PASSED"
for wb in "3 1" "2 1" "4 1" "3 2" "3 1"; do
  set -- $wb
  echo "== W=$1 B=$2"
  CUHE_SCHED_BATCH_WORKERS=$2 timeout 300 $L/test_prince_flow --threads 1 --sched $1 --no-round-checks --repeat 4 2>&1 | grep -E "Prince Encryption|FAILED" | awk '{printf "%s ", $3} END {print ""}'
done
export TMPDIR=/tmp CUHE_TRACE_MARK=1
cd /tmp
rm -rf /tmp/pa /tmp/ps
timeout 300 rocprofv3 --kernel-trace -d /tmp/pa -o s -- $R/$L/test_prince_arrays_cxx --no-round-checks --async 2>&1 | grep -E "Prince Encryption|FAILED"
python $R/tools/rocpd_summary.py --between k_probe_valu /tmp/pa/s_results.db 2>&1 | cut -c1-100,111-160 > $R/gpurun_out/r05_gaps_arrays2.txt
timeout 300 rocprofv3 --kernel-trace -d /tmp/ps -o s -- $R/$L/test_prince_flow --threads 1 --sched --no-round-checks --repeat 2 2>&1 | grep -E "Prince Encryption|FAILED"
python $R/tools/rocpd_summary.py --between k_probe_valu /tmp/ps/s_results.db 2>&1 | cut -c1-100,111-160 > $R/gpurun_out/r05_gaps_sched2.txt
cd $R
grep -n "^--\|idle\|gap " gpurun_out/r05_gaps_arrays2.txt | cut -c1-200
grep -n "^--\|idle gaps" gpurun_out/r05_gaps_sched2.txt | cut -c1-200
