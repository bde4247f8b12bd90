#!/bin/bash
export TMPDIR=/tmp
R=$PWD; out=$R/gpurun_out/r03e; mkdir -p $out
timeout 300 $R/cuhe_amd/lib/ow_ab 4096 10 > $out/ow_ab.txt 2>&1; cat $out/ow_ab.txt
cd /tmp
i=0
for set in "SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "GRBM_GUI_ACTIVE FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $set -d /tmp/pm_$i -o p -- $R/cuhe_amd/lib/ow_ab 1024 2 > /dev/null 2>&1
  python $R/tools/rocpd_summary.py /tmp/pm_$i/p_results.db 2>&1 | grep -E "^kernel|onewg|pass[12]w<16" | cut -c1-60,100-220 > $out/pmc_$i.txt
done
cat $out/pmc_3.txt $out/pmc_2.txt | grep -E "SIZE"
