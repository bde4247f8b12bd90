#!/bin/bash
# round 5, GPU run 4: the kept CRT rows (A/B), defaults as shipped, kernel trace of the default, C++ API tests
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
L=cuhe_amd/lib
export TMPDIR=/tmp
R=$PWD
{
echo "== API test (synchronous, then scheduled with the mirror check)"
timeout 600 $L/test_cuhe_api 2>&1 | grep -v "^ok" | tail -6
CUHE_SCHED=1 CUHE_SCHED_CHECK=1 timeout 600 $L/test_cuhe_api 2>&1 | grep -v "^ok" | tail -6
for keep in 1 0 1 0; do
  echo "== scheduled, defaults, CUHE_KEEP_CRT=$keep: 4 blocks"
  CUHE_SCHED_STATS=1 CUHE_KEEP_CRT=$keep timeout 300 $L/test_prince_flow --threads 1 --sched --no-round-checks --repeat 4 2>&1 | grep -E "Prince Encryption|batches:" | cut -c1-200
done
for keep in 1 0; do
  echo "== synchronous gates, 1 thread, CUHE_KEEP_CRT=$keep"
  CUHE_KEEP_CRT=$keep timeout 300 $L/test_prince_flow --threads 1 --no-round-checks 2>&1 | grep -E "Prince Encryption"
  echo "== asynchronous gates, 4 threads, CUHE_KEEP_CRT=$keep"
  CUHE_KEEP_CRT=$keep timeout 300 $L/test_prince_flow --threads 4 --async --no-round-checks 2>&1 | grep -E "Prince Encryption"
done
echo "== default, round checks + mirror check"
CUHE_SCHED_CHECK=1 timeout 300 $L/test_prince_flow --threads 1 --sched 2>&1 | grep -E "S-box layer|Prince Encryption|homomorphic PRINCE|PASSED|FAILED"
echo "== 8 virtual devices, round checks"
CUHE_SCHED_CHECK=1 CUHE_SCHED_STATS=1 timeout 300 $L/test_prince_flow --threads 1 --sched --devices 8 --virtual --repeat 2 2>&1 | grep -E "Prince Encryption|homomorphic PRINCE|PASSED|FAILED|batches:" | cut -c1-200
echo "== arrays client"
timeout 300 $L/test_prince_arrays_cxx --no-round-checks --async 2>&1 | tail -2
} > gpurun_out/r05_sched_run4.txt 2>&1
( cd /tmp && rm -rf /tmp/ps && CUHE_SCHED_STATS=1 CUHE_SCHED_TRACE=1 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/ps -o s -- $R/$L/test_prince_flow --threads 1 --sched --no-round-checks 2>&1 | grep -E "Prince Encryption|PASSED|batches:|scheduler:|device 0"; python $R/tools/rocpd_summary.py /tmp/ps/s_results.db 2>&1 | head -36 | cut -c1-90,112-175 ) > gpurun_out/r05_sched_trace2.txt 2>&1
cat gpurun_out/r05_sched_run4.txt; head -60 gpurun_out/r05_sched_trace2.txt
