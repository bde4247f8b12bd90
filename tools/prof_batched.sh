#!/bin/bash
# kernel trace of the batched multiply + relinearise call alone: tools/prof_batched.sh [batch] [ring]
export TMPDIR=/tmp
R=$PWD; B=${1:-32}; RING=${2:-2^15}
mkdir -p gpurun_out/r02
cd /tmp; rm -rf /tmp/pb
timeout 150 rocprofv3 --kernel-trace --stats -d /tmp/pb -o s -- python $R/tools/trace_batched.py $B 10 $RING 2>&1 | grep "per ciphertext"
python $R/tools/rocpd_summary.py /tmp/pb/s_results.db 2>&1 | head -20 | cut -c1-84,112-200 | tee $R/gpurun_out/r02/batched_trace_b$B.txt
