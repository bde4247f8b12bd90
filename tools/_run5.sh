cd $GRAFT_REPO_ROOT
L=cuhe_amd/lib
export CUHE_SCHED_STATS=1
{
for rep in 1 2; do for b in 64 32; do for w in 3 4; do
  echo "== PRINCE sched $w workers, batches of up to $b"
  CUHE_SCHED_BATCH=$b timeout 900 $L/test_prince_flow --threads 1 --sched $w --no-round-checks 2>&1 | tail -8 | grep "Prince Enc\|batches\|wrong\|FAILED"
done; done; done
echo "== compare (bench leg)"
timeout 900 $L/test_prince_flow --threads 1 --no-round-checks --compare 2>&1 | grep "Prince Enc\|right\|wrong"
} > gpurun_out/run5.txt 2>&1
cat gpurun_out/run5.txt
