#!/bin/bash
export TMPDIR=/tmp
R=$PWD; out=$R/gpurun_out/r03n; mkdir -p $out
cd /tmp
for ring in "2^16" "2^15"; do
  rm -rf /tmp/pf_b
  timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pf_b -o s -- python $R/tools/trace_batched.py 32 10 $ring > $out/trace_$ring.txt 2>&1
  python $R/tools/rocpd_summary.py /tmp/pf_b/s_results.db 2>&1 | head -16 | cut -c1-84,112-200 >> $out/trace_$ring.txt
  cat $out/trace_$ring.txt | tail -17
done
