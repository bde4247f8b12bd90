#!/bin/bash
# A/B of the low-latency transform kernels on whole programs: the metric's batch with pass 2 alone switched
# (CUHE_LL2_ROWS), and the gate-by-gate homomorphic PRINCE with the kernels off / on (CUHE_LL_ROWS).
for LL2 in -1 1000000; do CUHE_LL2_ROWS=$LL2 timeout 200 python bench.py --steps 100 --warmup 10 --no-mulrelin --no-cpu --no-prince 2>/dev/null | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.read()); r = j['roofline']
print('metric batch, CUHE_LL2_ROWS=$LL2: NTT/s', j['value'], 'frac', r['frac'], 'pass1 ms', r.get('pass1_ms_per_batch'), 'pass2 ms', r.get('pass2_ms_per_batch'))"; done
make -C cuhe_amd/cxx -s test > /dev/null 2>&1
for LL in 0 24; do for F in "--threads 8" "--threads 1" "--threads 4 --async"; do echo "CUHE_LL_ROWS=$LL  $F:"; CUHE_LL_ROWS=$LL timeout 200 cuhe_amd/lib/test_prince_flow $F --no-round-checks 2>&1 | grep -E "Prince Encryption|PASSED|FAILED"; done; done
