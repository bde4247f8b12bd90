#!/bin/bash
# where the device sits idle (tools/rocpd_summary.py --between: idle gaps by size and by the kernel that follows): the arrays client, the scheduled
# gate-by-gate client (second block) and the batched multiply + relinearise call on both config-4 rings
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
R=$PWD; L=cuhe_amd/lib
export TMPDIR=/tmp CUHE_TRACE_MARK=1
cd /tmp
rm -rf /tmp/pa /tmp/ps /tmp/pb15 /tmp/pb16
timeout 300 rocprofv3 --kernel-trace -d /tmp/pa -o s -- $R/$L/test_prince_arrays_cxx --no-round-checks --async 2>&1 | grep -E "Prince Encryption|FAILED"
python $R/tools/rocpd_summary.py --between k_probe_valu /tmp/pa/s_results.db 2>&1 | cut -c1-100,111-160 > $R/gpurun_out/r05_gaps_arrays.txt
timeout 300 rocprofv3 --kernel-trace -d /tmp/ps -o s -- $R/$L/test_prince_flow --threads 1 --sched --no-round-checks --repeat 2 2>&1 | grep -E "Prince Encryption|FAILED"
python $R/tools/rocpd_summary.py --between k_probe_valu /tmp/ps/s_results.db 2>&1 | cut -c1-100,111-160 > $R/gpurun_out/r05_gaps_sched.txt
for ring in 2^15 2^16; do
  timeout 300 rocprofv3 --kernel-trace -d /tmp/pb$ring -o s -- python $R/tools/trace_batched.py 32 10 $ring 2>&1 | grep "ms per"
  python $R/tools/rocpd_summary.py --between k_probe_valu /tmp/pb$ring/s_results.db 2>&1 | cut -c1-100,111-160 > $R/gpurun_out/r05_gaps_batched_$ring.txt
done
cd $R
grep -n "^--\|idle" gpurun_out/r05_gaps_arrays.txt gpurun_out/r05_gaps_batched_*.txt | cut -c1-220
