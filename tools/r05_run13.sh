#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
R=$PWD; L=cuhe_amd/lib
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "separately_owned or list_block" 2>&1 | tail -6
for v in 1 0 1; do
  echo "== CUHE_SCHED_LISTS=$v"
  CUHE_SCHED_LISTS=$v timeout 300 $L/test_prince_flow --threads 1 --sched --no-round-checks --repeat 4 2>&1 | grep -E "Prince Encryption|FAILED" | awk '{printf "%s ", $3} END {print ""}'
done
timeout 900 python -m pytest tests/test_gpu_cxx_api.py -q -x -k "scheduled" 2>&1 | tail -3
