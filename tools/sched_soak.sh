#!/bin/bash
# the soak behind "scheduled gates are the default".  Every tests/cxx program and the DHS flow on three rings, REPS
# times each, with scheduled gates (the library default: no CUHE_SCHED in the environment) and the mirror check on; test_sched_soak's
# conditions; the allocation-failure injection; test_cuhe_api with 1 / 2 / 3 workers per device.  One line per program: runs, failures,
# seconds.  Output: gpurun_out/sched_soak.txt (copied to profiles/r06_sched_soak.txt)
REPS=${1:-20}
OUT=gpurun_out/sched_soak.txt
L=cuhe_amd/lib
mkdir -p gpurun_out; : > $OUT
make -C cuhe_amd/cxx -s test >> $OUT 2>&1
unset CUHE_SCHED
export CUHE_SCHED_CHECK=1 CUHE_SCHED_WATCHDOG_S=30
now() { python3 -c 'import time; print(time.time())'; }
run() {   # name reps expected-substring cmd...
  local name=$1 reps=$2 want=$3; shift 3
  local bad=0 t0=$(now)
  for i in $(seq 1 $reps); do
    timeout 900 "$@" > gpurun_out/soak_one.log 2>&1; rc=$?
    if [ $rc -ne 0 ] || ! grep -q "$want" gpurun_out/soak_one.log || grep -q "wrong\|watchdog\|FAIL" gpurun_out/soak_one.log; then
      bad=$((bad+1)); echo "--- $name run $i: exit $rc" >> $OUT; tail -15 gpurun_out/soak_one.log >> $OUT
    fi
  done
  python3 -c "import time,sys; print('%-58s %3d runs, %d failed, %.1f s per run' % (sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), (time.time()-float(sys.argv[4]))/int(sys.argv[2])))" "$name" $reps $bad $t0 >> $OUT
}
echo "scheduled gates = the library default (CUHE_SCHED unset), CUHE_SCHED_CHECK=1, watchdog 30 s" >> $OUT
run "test_dhs_flow toy1155 (+ checkKeys)"            $REPS "ALL PASSED" $L/test_dhs_flow 3 2 8 40 20 1155
run "test_dhs_flow dhs_simple 8191 (+ checkKeys)"    $REPS "ALL PASSED" $L/test_dhs_flow 5 2 1 61 20 8191
run "test_dhs_flow x^16384+1 negacyclic (+ checkKeys)" $REPS "ALL PASSED" $L/test_dhs_flow 3 2 16 48 24 32768
run "test_sched_soak 3 (five conditions x 3)"        $REPS "ALL PASSED" $L/test_sched_soak 3
run "test_prince_flow --default (1 client thread)"   5 "ALL PASSED" $L/test_prince_flow --threads 1 --default --no-round-checks
run "test_prince_flow --default, round states"       2 "ALL PASSED" $L/test_prince_flow --threads 1 --default
run "test_prince_flow --default 8 client threads"    3 "ALL PASSED" $L/test_prince_flow --threads 8 --default --no-round-checks
run "test_prince_flow --default 3 virtual devices"   3 "ALL PASSED" $L/test_prince_flow --threads 1 --default --no-round-checks --devices 3 --virtual
run "test_prince_flow --zzx-state --default 8 client threads" 3 "ALL PASSED" $L/test_prince_flow --threads 8 --zzx-state --default --no-round-checks
run "test_prince_flow --zzx-state --default 1 thread, round states" 1 "ALL PASSED" $L/test_prince_flow --threads 1 --zzx-state --default
run "test_prince_flow --zzx-state --default 6 threads 3 virtual devices" 2 "ALL PASSED" $L/test_prince_flow --threads 6 --zzx-state --default --no-round-checks --devices 3 --virtual
run "test_multi_device (3 virtual devices)"          5 "ALL PASSED" $L/test_multi_device
export CUHE_SCHED=1
run "test_prince_batched (C-ABI arrays), CUHE_SCHED=1" 3 "ALL PASSED" $L/test_prince_batched
run "test_prince_arrays_cxx, CUHE_SCHED=1"           3 "ALL PASSED" $L/test_prince_arrays_cxx
run "test_prince_arrays_cxx --async, CUHE_SCHED=1"   3 "ALL PASSED" $L/test_prince_arrays_cxx --async
unset CUHE_SCHED
for T in 1 2 3; do
  CUHE_SCHED_THREADS=$T run "test_cuhe_api, $T worker(s) per device" 2 "ALL PASSED" $L/test_cuhe_api
done
for N in 0 1 7 40 150 600; do
  for S in unset 0; do
    if [ $S = 0 ]; then export CUHE_SCHED=0; else unset CUHE_SCHED; fi
    t0=$(now); timeout 120 $L/test_sched_soak allocfail $N > gpurun_out/soak_one.log 2>&1; rc=$?
    echo "allocation failure injected at allocation $N, CUHE_SCHED $S: exit $rc ($(grep -c 'cuheSafeCall() failed' gpurun_out/soak_one.log) cuheSafeCall message(s)), $(python3 -c "import time,sys; print('%.1f' % (time.time()-float(sys.argv[1])))" $t0) s" >> $OUT
  done
done
unset CUHE_SCHED
cat $OUT
