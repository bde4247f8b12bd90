#!/bin/bash
# duration of the forward transform pair at small row counts, throughput kernels (CUHE_LL_ROWS=0) against the
# low-latency kernels (CUHE_LL_ROWS large): bench.py --len L --batch B, hipEvent timing of the two kernels
for L in 32768 65536; do for B in 1 8 24 48 96 192; do for LL in 0 1000000; do
  CUHE_LL_ROWS=$LL python bench.py --len $L --batch $B --steps 200 --warmup 20 --no-mulrelin --no-cpu --no-prince 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        j = json.loads(line); r = j['roofline']
        print('L $L batch %4d ll %d: pass1 %.1f us  pass2 %.1f us  pair %.1f us  (%.0f NTT/s)' % ($B, 1 if $LL else 0, 1e3 * r['pass1_ms_per_batch'], 1e3 * r['pass2_ms_per_batch'], 1e3 * r['pipelined_ms_per_batch'], j['value']))
"
done; done; done
