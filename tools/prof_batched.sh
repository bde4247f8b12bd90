#!/bin/bash
# kernel trace of bench.py's batched multiply + relinearise leg at batch $1 (default 32)
export TMPDIR=/tmp
R=$PWD; B=${1:-32}
mkdir -p gpurun_out/r02
cd /tmp; rm -rf /tmp/pb
timeout 120 rocprofv3 --kernel-trace --stats -d /tmp/pb -o s -- python $R/bench.py --steps 3 --warmup 1 --no-cpu --no-prince --one-ring --relin-batch $B > /dev/null 2>&1
python $R/tools/rocpd_summary.py /tmp/pb/s_results.db 2>&1 | head -22 | cut -c1-84,112-200 | tee $R/gpurun_out/r02/batched_trace_b$B.txt
