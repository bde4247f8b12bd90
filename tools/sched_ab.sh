#!/bin/bash
# scheduled gates vs the reference's synchronous gates, one client thread: PRINCE gate by gate, the
# DHS flow and the API test under CUHE_SCHED=1 with the mirror check on.  Output: gpurun_out/sched_ab.txt
cd "$(dirname "$0")/.." || exit 1
out=gpurun_out/sched_ab.txt
mkdir -p gpurun_out
L=cuhe_amd/lib
export CUHE_SCHED_STATS=1
{
echo "== API test, scheduled (CUHE_SCHED=1 CUHE_SCHED_CHECK=1)"
CUHE_SCHED=1 CUHE_SCHED_CHECK=1 timeout 600 $L/test_cuhe_api 2>&1 | grep -v "^ok" | tail -12
echo "== PRINCE, 1 thread, synchronous gates (the reference's pattern)"
timeout 900 $L/test_prince_flow --threads 1 --no-round-checks 2>&1 | tail -2
for loc in 1 0; do for w in 3 4 6 8; do
  echo "== PRINCE, 1 thread, scheduled gates, $w workers, local queues $loc"
  CUHE_SCHED_LOCAL=$loc timeout 900 $L/test_prince_flow --threads 1 --sched $w --no-round-checks 2>&1 | tail -5 | grep -v circuit
done; done
echo "== PRINCE, 1 thread, scheduled gates, 6 workers, GPU_MAX_HW_QUEUES=2"
GPU_MAX_HW_QUEUES=2 timeout 900 $L/test_prince_flow --threads 1 --sched 6 --no-round-checks 2>&1 | tail -5 | grep -v circuit
echo "== PRINCE, 4 threads asynchronous (r03 reference point)"
timeout 900 $L/test_prince_flow --threads 4 --async --no-round-checks 2>&1 | tail -3
} > $out 2>&1
cat $out
