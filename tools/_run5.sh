cd $GRAFT_REPO_ROOT
L=cuhe_amd/lib
export CUHE_SCHED_STATS=1
{
echo "== API test, scheduled + batching"
CUHE_SCHED=1 CUHE_SCHED_CHECK=1 timeout 600 $L/test_cuhe_api 2>&1 | grep -v "^ok" | tail -4
for rep in 1 2 3; do for w in 2 3 4; do
  echo "== PRINCE sched $w workers"
  timeout 900 $L/test_prince_flow --threads 1 --sched $w --no-round-checks 2>&1 | tail -8 | grep "Prince Enc\|wrong\|FAILED"
done; done
echo "== PRINCE sched 3 workers, round checks + mirror check"
CUHE_SCHED_CHECK=1 timeout 900 $L/test_prince_flow --threads 1 --sched 3 2>&1 | grep "Prince Enc\|batches\|wrong\|FAILED\|PASSED" 
echo "== PRINCE arrays client"
timeout 900 $L/test_prince_arrays_cxx --no-round-checks --async 2>&1 | grep "Prince\|PASSED\|FAILED" | tail -3
} > gpurun_out/run5.txt 2>&1
cat gpurun_out/run5.txt
