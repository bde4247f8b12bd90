#!/bin/bash
export TMPDIR=/tmp
R=$PWD; out=$R/gpurun_out/r03p; mkdir -p $out
timeout 300 $R/cuhe_amd/lib/ow_ab 4096 10 > $out/ow_ab.txt 2>&1; grep -E "65536" $out/ow_ab.txt
cd /tmp
for set in "WRITE_SIZE" "FETCH_SIZE"; do
  timeout 200 rocprofv3 --kernel-trace --pmc $set -d /tmp/pm_$set -o p -- $R/cuhe_amd/lib/ow_ab 1024 2 > /dev/null 2>&1
  python $R/tools/rocpd_summary.py /tmp/pm_$set/p_results.db 2>&1 | grep -E "SIZE" | grep -E "stream|pass[12]w<16" | cut -c1-60,100-220 | tee -a $out/pmc.txt
done
