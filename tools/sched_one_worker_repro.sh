#!/bin/bash
# tests/cxx/test_cuhe_api stalled with ONE scheduler worker per device on the GPU box.  Runs it with CUHE_SCHED_THREADS=$1
# (default 1) $2 times under the watchdog (cuhe_amd/cxx/Scheduler.cpp: dumpState) as a child of rocgdb; a run still alive after 45 s gets
# SIGUSR1, which stops it inside the debugger, and the backtrace of every thread is taken.  Output: gpurun_out/sched_one_worker.txt
T=${1:-1}; N=${2:-2}
OUT=gpurun_out/sched_one_worker.txt
mkdir -p gpurun_out; : > $OUT
make -C cuhe_amd/cxx -s test >> $OUT 2>&1
for i in $(seq 1 $N); do
  echo "=== run $i: CUHE_SCHED=1 CUHE_SCHED_THREADS=$T" >> $OUT
  CUHE_SCHED=1 CUHE_SCHED_CHECK=1 CUHE_SCHED_THREADS=$T CUHE_SCHED_WATCHDOG_S=15 stdbuf -oL -eL \
    /opt/rocm/bin/rocgdb -batch -ex "handle SIGUSR1 stop print nopass" -ex run -ex "thread apply all bt 30" -ex kill --args cuhe_amd/lib/test_cuhe_api > gpurun_out/one_worker_run_$i.log 2>&1 &
  gpid=$!
  for s in $(seq 1 45); do sleep 1; kill -0 $gpid 2>/dev/null || break; done
  if kill -0 $gpid 2>/dev/null; then
    child=$(pgrep -P $gpid | head -1)
    echo "--- still running after 45 s: SIGUSR1 to the inferior $child" >> $OUT
    kill -USR1 $child
    for s in $(seq 1 90); do sleep 1; kill -0 $gpid 2>/dev/null || break; done
    kill -9 $gpid 2>/dev/null
  fi
  wait $gpid 2>/dev/null
  grep -v "^\[New Thread\|^\[Thread .* exited\|^ok: " gpurun_out/one_worker_run_$i.log | tail -250 >> $OUT
done
tail -300 $OUT
