#!/bin/bash
# round 6, first full pass on the GPU: allocation-failure injection after the worker-exit fix, the whole -m gpu suite (scheduled gates are
# the default now), the RowRebase kernel-argument A/B
mkdir -p gpurun_out
L=cuhe_amd/lib
OUT=gpurun_out/r06_allocfail.txt; : > $OUT
for N in 0 1 2 3 7 20 40 80 150 200; do
  for S in unset 0; do
    if [ $S = 0 ]; then export CUHE_SCHED=0; else unset CUHE_SCHED; fi
    s=$(date +%s.%N); timeout 120 $L/test_sched_soak allocfail $N > gpurun_out/soak_one.log 2>&1; rc=$?
    echo "allocation failure injected at allocation $N, CUHE_SCHED $S: exit $rc ($(grep -c 'cuheSafeCall() failed' gpurun_out/soak_one.log) cuheSafeCall message(s)), $(python3 -c "import time,sys; print('%.1f' % (time.time()-float(sys.argv[1])))" $s) s" >> $OUT
  done
done
unset CUHE_SCHED
cat $OUT
python tools/rebase_arg_ab.py > gpurun_out/r06_rebase_arg_ab.txt 2>&1; cat gpurun_out/r06_rebase_arg_ab.txt
timeout 1500 python -m pytest tests -q -m gpu -x --durations=30 > gpurun_out/r06_pytest_gpu.txt 2>&1; tail -45 gpurun_out/r06_pytest_gpu.txt
