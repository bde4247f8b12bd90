#!/bin/bash
# round 5, GPU run 1: the GPU suite on the rewritten scheduler / list kernels / exchange paths, then the scheduled PRINCE under
# every batch policy (stats + group-size trace), a kernel trace of the default, 8 virtual devices, and one bench line with the limiter.
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
L=cuhe_amd/lib
export TMPDIR=/tmp
( timeout 900 python -m pytest tests -m gpu -x -q --durations=8 2>&1 | tail -25 ) > gpurun_out/r05_pytest_gpu.txt
{
for rep in 1 2; do
for cfg in "0 0" "0 1" "1 1" "2 1"; do
  set -- $cfg
  echo "== policy $1 lists $2 (3 workers)"
  CUHE_SCHED_STATS=1 CUHE_SCHED_TRACE=$([ $rep = 1 ] && echo 1 || echo 0) CUHE_SCHED_POLICY=$1 CUHE_SCHED_LISTS=$2 timeout 300 $L/test_prince_flow --threads 1 --sched 3 --no-round-checks 2>&1 | grep -E "Prince Encryption|recorded|batches:|scheduler:|device 0|homomorphic PRINCE"
done; done
for w in 2 4 5; do for pol in 1 2; do
  echo "== policy $pol, $w workers"
  CUHE_SCHED_STATS=1 CUHE_SCHED_POLICY=$pol timeout 300 $L/test_prince_flow --threads 1 --sched $w --no-round-checks 2>&1 | grep -E "Prince Encryption|batches:"
done; done
echo "== policy 1, quiet 0"
CUHE_SCHED_STATS=1 CUHE_SCHED_POLICY=1 CUHE_SCHED_QUIET_US=0 timeout 300 $L/test_prince_flow --threads 1 --sched 3 --no-round-checks 2>&1 | grep -E "Prince Encryption|batches:"
echo "== default, round checks + mirror check"
CUHE_SCHED_CHECK=1 timeout 300 $L/test_prince_flow --threads 1 --sched 3 2>&1 | grep -E "S-box layer|Prince Encryption|homomorphic PRINCE|PASSED|FAILED"
echo "== 8 virtual devices, scheduled / 1 device"
CUHE_SCHED_STATS=1 timeout 300 $L/test_prince_flow --threads 1 --sched 3 --no-round-checks --devices 8 --virtual 2>&1 | grep -E "Prince Encryption|batches:|scheduler:|homomorphic PRINCE"
CUHE_SCHED_STATS=1 timeout 300 $L/test_prince_flow --threads 1 --sched 3 --no-round-checks --devices 3 --virtual 2>&1 | grep -E "Prince Encryption|batches:|homomorphic PRINCE"
echo "== arrays client (reference point)"
timeout 300 $L/test_prince_arrays_cxx --no-round-checks --async 2>&1 | tail -2
} > gpurun_out/r05_sched_ab.txt 2>&1
R=$PWD
( cd /tmp && rm -rf /tmp/ps && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/ps -o s -- $R/$L/test_prince_flow --threads 1 --sched 3 --no-round-checks 2>&1 | grep -E "Prince Encryption|PASSED"; python $R/tools/rocpd_summary.py /tmp/ps/s_results.db 2>&1 | head -45 | cut -c1-90,112-175 ) > gpurun_out/r05_sched_trace.txt 2>&1
( timeout 600 python bench.py 2>&1 | tail -3 ) > gpurun_out/r05_bench1.txt
tail -3 gpurun_out/r05_pytest_gpu.txt; cat gpurun_out/r05_sched_ab.txt | head -150; tail -2 gpurun_out/r05_bench1.txt | cut -c1-3000
