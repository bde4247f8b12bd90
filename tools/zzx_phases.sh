#!/bin/bash
# The reference example's literal client structure (test_prince_flow --zzx-state: ZZX state on the host between S-boxes, Prince.cu:188-322) by client
# threads and gate mode, with the client threads' time by phase.  A/B switches: CUHE_CLIENT_STAGING=0 = host halves of the ZZX conversions on the

# (the bench's gate-by-gate leg) on the same switches: it must not move.  Output: gpurun_out/zzx_phases.txt
L=cuhe_amd/lib; OUT=gpurun_out/zzx_phases.txt; mkdir -p gpurun_out; : > $OUT
make -C cuhe_amd/cxx -s test >> $OUT 2>&1
for t in ${THREADS:-1 8 16}; do
for v in "CUHE_SCHED=0" "X=default" "CUHE_SCHED_LATENCY=0" "CUHE_CLIENT_STAGING=0" "CUHE_SCHED_LATENCY=0 CUHE_CLIENT_STAGING=0" $EXTRA; do
  echo "== zzx-state, $t client thread(s), $v" >> $OUT
  env $v timeout 300 $L/test_prince_flow --threads $t --zzx-state --default --no-round-checks --repeat 3 2>&1 | grep -E "Prince Enc|client-thread|wrong|FAILED" | sed -E 's/ on 1 device.*//' >> $OUT
done; done
for v in "X=default" "CUHE_SCHED_LATENCY=0"; do
  echo "== device-resident state, 1 client thread, $v" >> $OUT
  env $v timeout 300 $L/test_prince_flow --threads 1 --default --no-round-checks --repeat 4 2>&1 | grep -E "Prince Enc|wrong|FAILED" | sed -E 's/ on 1 device.*//' >> $OUT
done
cat $OUT
