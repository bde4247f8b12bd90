#!/bin/bash
# A/B of the low-latency transform kernels on a whole program: the gate-by-gate homomorphic PRINCE with the kernels
# off / on (CUHE_LL_ROWS).  (profiles/r02_low_latency_ab.txt also holds the metric's batch with pass 2 alone switched to
# the low-latency form -- 1.74 vs 1.49 ms per 512 transforms -- measured with a knob that has since been removed.)
make -C cuhe_amd/cxx -s test > /dev/null 2>&1
for LL in 0 24; do for F in "--threads 8" "--threads 1" "--threads 4 --async"; do echo "CUHE_LL_ROWS=$LL  $F:"; CUHE_LL_ROWS=$LL timeout 200 cuhe_amd/lib/test_prince_flow $F --no-round-checks 2>&1 | grep -E "Prince Encryption|PASSED|FAILED"; done; done
