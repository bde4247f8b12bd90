#!/bin/bash
# release-only tasks + wait de-duplication in the gate scheduler: parity of the scheduled PRINCE modes, then timing (4 blocks per process, twice),
# then the kernel trace cut to the timed part
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
R=$PWD; L=cuhe_amd/lib
timeout 900 python -m pytest tests/test_gpu_cxx_api.py -q -x -k "scheduled" 2>&1 | tail -5
for i in 1 2; do
  CUHE_SCHED_STATS=1 timeout 300 $L/test_prince_flow --threads 1 --sched --no-round-checks --repeat 4 2>&1 | grep -E "Prince Encryption|batches:|scheduler:|task blocks|PASSED|FAILED"
done
timeout 300 $L/test_prince_arrays_cxx --no-round-checks --async 2>&1 | grep -E "Prince Encryption"
export TMPDIR=/tmp CUHE_TRACE_MARK=1
cd /tmp; rm -rf /tmp/ps
timeout 300 rocprofv3 --kernel-trace -d /tmp/ps -o s -- $R/$L/test_prince_flow --threads 1 --sched --no-round-checks --repeat 2 2>&1 | grep -E "Prince Encryption|PASSED|FAILED"
python $R/tools/rocpd_summary.py --between k_probe_valu /tmp/ps/s_results.db 2>&1 | cut -c1-100,111-160 > $R/gpurun_out/r05_cut_sched2.txt
grep -n "^--\|^==" $R/gpurun_out/r05_cut_sched2.txt
