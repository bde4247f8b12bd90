#!/bin/bash
# kernel traces of ONE PRINCE block cut to the timed part (CUHE_TRACE_MARK=1, tools/rocpd_summary.py --between): the arrays client
# against the scheduled gate-by-gate client (second block = warm scratch), so that the two kernel mixes can be read side by side
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
R=$PWD; L=cuhe_amd/lib
export TMPDIR=/tmp CUHE_TRACE_MARK=1
cd /tmp
rm -rf /tmp/pa /tmp/ps
timeout 300 rocprofv3 --kernel-trace -d /tmp/pa -o s -- $R/$L/test_prince_arrays_cxx --no-round-checks --async 2>&1 | grep -E "Prince Encryption|PASSED|FAILED"
python $R/tools/rocpd_summary.py --between k_probe_valu /tmp/pa/s_results.db 2>&1 | cut -c1-100,111-160 > $R/gpurun_out/r05_cut_arrays.txt
CUHE_SCHED_STATS=1 timeout 300 rocprofv3 --kernel-trace -d /tmp/ps -o s -- $R/$L/test_prince_flow --threads 1 --sched --no-round-checks --repeat 2 2>&1 | grep -E "Prince Encryption|PASSED|FAILED|batches:"
python $R/tools/rocpd_summary.py --between k_probe_valu /tmp/ps/s_results.db 2>&1 | cut -c1-100,111-160 > $R/gpurun_out/r05_cut_sched.txt
cd $R
grep -n "^--\|^==" gpurun_out/r05_cut_arrays.txt gpurun_out/r05_cut_sched.txt
