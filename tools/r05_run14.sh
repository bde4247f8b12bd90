#!/bin/bash
# bisect the scheduled failure of test_cuhe_api (Operations.h section): repeat, with the new paths switched off one at a time
cd "$(dirname "$0")/.." || exit 1
L=cuhe_amd/lib
run() { echo "== $*"; for i in 1 2 3 4; do env "$@" CUHE_SCHED=1 CUHE_SCHED_CHECK=1 timeout 120 $L/test_cuhe_api 2>&1 | grep -E "^FAIL|PASSED|FAILED \(" | tr '\n' ' '; echo; done; }
run X=1
run CUHE_ROW_LISTS=0
run CUHE_SCHED_LISTS=0
run CUHE_SCHED_BATCH=0
run CUHE_SCHED_THREADS=1
echo "== synchronous"; for i in 1 2; do timeout 120 $L/test_cuhe_api 2>&1 | grep -E "^FAIL|PASSED|FAILED \(" | tr '\n' ' '; echo; done
