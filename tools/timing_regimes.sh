#!/bin/bash
# the scheduled-gate clients under other TIMING regimes than the default: everything on the far socket, on 2 CPUs only, no spinning, one worker, latency mode
# off -- races that hide behind the usual interleavings show up as wrong results, watchdog reports or hangs here.  Output: gpurun_out/timing_regimes.txt
L=cuhe_amd/lib; OUT=gpurun_out/timing_regimes.txt; mkdir -p gpurun_out; : > $OUT
make -C cuhe_amd/cxx -s test >> $OUT 2>&1
export CUHE_SCHED_CHECK=1 CUHE_SCHED_WATCHDOG_S=30
FAR=64-127
one() {   # label, env..., --, command
  local label=$1; shift
  local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  local t0=$(date +%s.%N)
  env "${envs[@]}" timeout 600 "$@" > gpurun_out/regime_one.log 2>&1; local rc=$?
  local ok=ok; if [ $rc -ne 0 ] || ! grep -q "ALL PASSED" gpurun_out/regime_one.log || grep -q "wrong\|watchdog\|FAIL" gpurun_out/regime_one.log; then ok=FAILED; tail -12 gpurun_out/regime_one.log >> $OUT; fi
  printf "%-78s exit %d  %s  %s s  %s\n" "$label" $rc $ok $(python3 -c "import time,sys; print(\"%.1f\" % (time.time()-float(sys.argv[1])))" $t0) "$(grep -E 'Prince Encryption' gpurun_out/regime_one.log | sed -E 's/Prince Encryption: ([0-9.]+) s.*/\1/' | tr '\n' ' ')" >> $OUT
}
for regime in "far-socket:taskset -c $FAR" "two-cpus:taskset -c 0,1" "far-socket-unpinned-workers:taskset -c $FAR"; do
  name=${regime%%:*}; pre=${regime#*:}
  extra=X=1; [ $name = far-socket-unpinned-workers ] && extra=CUHE_SCHED_PIN=0
  one "$name: test_sched_soak 3"                        $extra -- $pre $L/test_sched_soak 3
  one "$name: prince, device-resident, 1 thread"        $extra -- $pre $L/test_prince_flow --threads 1 --default --no-round-checks --repeat 2
  one "$name: prince, literal client, 8 threads"        $extra -- $pre $L/test_prince_flow --threads 8 --zzx-state --default --no-round-checks --repeat 2
  one "$name: dhs flow x^16384+1"                       $extra -- $pre $L/test_dhs_flow 3 2 16 48 24 32768
done
for v in "CUHE_SCHED_SPIN_US=0" "CUHE_SCHED_THREADS=1" "CUHE_SCHED_LATENCY=0" "CUHE_SCHED_THREADS=8" "CUHE_SCHED_BATCH_WORKERS=0"; do
  one "$v: test_sched_soak 3"                           $v -- $L/test_sched_soak 3
  one "$v: prince, literal client, 8 threads"           $v -- $L/test_prince_flow --threads 8 --zzx-state --default --no-round-checks --repeat 2
  one "$v: prince, device-resident, round states"       $v -- $L/test_prince_flow --threads 1 --default
done
cat $OUT
