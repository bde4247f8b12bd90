#!/bin/bash
export TMPDIR=/tmp
R=$PWD; out=$R/gpurun_out/r03l; mkdir -p $out
for m in 0 1 2; do
  CUHE_ONEWG=$m CUHE_ONEWG64=$([ $m = 2 ] && echo 1 || echo 0) timeout 300 python bench.py --perf-table $out/perf_table_onewg$m.txt > /dev/null 2>&1
  echo "== CUHE_ONEWG=$m"; cat $out/perf_table_onewg$m.txt
done
