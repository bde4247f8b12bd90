#!/bin/bash
export TMPDIR=/tmp
R=$PWD; out=$R/gpurun_out/r03c; mkdir -p $out
timeout 200 $R/tools/ubench_onewg 2048 5 > $out/ubench_onewg.txt 2>&1; cat $out/ubench_onewg.txt
timeout 300 $R/cuhe_amd/lib/ow_ab 4096 10 > $out/ow_ab.txt 2>&1; grep -v mismatch $out/ow_ab.txt; grep -c identical $out/ow_ab.txt
