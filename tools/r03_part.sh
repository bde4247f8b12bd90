#!/bin/bash
export TMPDIR=/tmp
R=$PWD; out=$R/gpurun_out/r03m; mkdir -p $out
for s in 0 4 6 8 12; do
  for ring in "2^15" "2^16"; do
    CUHE_RELIN_PARTITION=$s timeout 300 python tools/trace_batched.py 32 10 $ring 2>&1 | tail -1 | sed "s/^/partition $s: /"
  done
done | tee $out/partition_sweep.txt
CUHE_RELIN_PARTITION=8 timeout 300 python tools/trace_batched.py 64 6 "2^15" 2>&1 | tail -1 | sed "s/^/partition 8 batch 64: /" | tee -a $out/partition_sweep.txt
CUHE_RELIN_PARTITION=0 timeout 300 python tools/trace_batched.py 64 6 "2^15" 2>&1 | tail -1 | sed "s/^/partition 0 batch 64: /" | tee -a $out/partition_sweep.txt
