#!/bin/bash
# wall time of tests/cxx/test_cuhe_api under scheduled gates with 1, 2 and 3 workers per device (and synchronous): the round-5 "stall with
# one worker" against the host-side check of the program (schoolbook ZZX products of 16384 coefficients on the fallback big integer)
OUT=gpurun_out/sched_one_worker_time.txt
mkdir -p gpurun_out; : > $OUT
make -C cuhe_amd/cxx -s test >> $OUT 2>&1
for T in 1 3 1 2 0; do
  s=$(date +%s.%N)
  if [ $T = 0 ]; then timeout 900 cuhe_amd/lib/test_cuhe_api > gpurun_out/t.log 2>&1; rc=$?
  else CUHE_SCHED=1 CUHE_SCHED_CHECK=1 CUHE_SCHED_THREADS=$T CUHE_SCHED_WATCHDOG_S=20 CUHE_SCHED_STATS=1 timeout 900 cuhe_amd/lib/test_cuhe_api > gpurun_out/t.log 2>&1; rc=$?; fi
  e=$(date +%s.%N)
  echo "workers per device $T (0 = synchronous gates): exit $rc, $(echo "$e - $s" | bc) s, $(grep -c '^ok:' gpurun_out/t.log) checks ok, $(grep -c '^FAIL' gpurun_out/t.log) failed; $(grep -i 'watchdog' gpurun_out/t.log | head -1)" >> $OUT
  grep "^scheduler:" gpurun_out/t.log | tail -1 >> $OUT
done
cat $OUT
