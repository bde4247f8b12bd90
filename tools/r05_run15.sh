#!/bin/bash
# which change makes the scheduled run of test_cuhe_api fail (Operations.h section)?  3 runs per variant, 40 s limit each
cd "$(dirname "$0")/.." || exit 1
L=$PWD/cuhe_amd/lib
run() { echo "== $*"; for i in 1 2 3; do env "$@" CUHE_SCHED=1 CUHE_SCHED_CHECK=1 timeout 40 $L/test_cuhe_api 2>&1 | grep -E "^FAIL|PASSED|FAILED \(" | cut -c1-60 | tr '\n' ' '; echo; done; }
run X=1
run LD_LIBRARY_PATH=$L/alt_morning
run CUHE_SCHED_RELEASE_ONLY=0
run CUHE_SCHED_WAIT_DEDUP=0
run CUHE_SCHED_RELEASE_ONLY=0 CUHE_SCHED_WAIT_DEDUP=0
