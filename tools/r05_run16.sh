#!/bin/bash
# the scheduled run of test_cuhe_api with the by-hand section's rows cleared: three processes at once (150 s limit each)
cd "$(dirname "$0")/.." || exit 1
L=$PWD/cuhe_amd/lib
for i in 1 2 3; do ( CUHE_SCHED=1 CUHE_SCHED_CHECK=1 timeout 150 $L/test_cuhe_api 2>&1 | grep -E "^FAIL|PASSED|FAILED \(" | cut -c1-70 | tr '\n' ' '; echo "[run $i]" ) & done
wait
