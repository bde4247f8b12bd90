#!/bin/bash
# how the pass-1 -> pass-2 slab size (transforms per launch pair) changes each pass's time
for c in 128 256 512 1024; do for ov in 0; do
  python bench.py --steps 10 --warmup 3 --no-mulrelin --no-cpu --chunk $c --overlap $ov 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        j = json.loads(line); r = j['roofline']
        print('chunk $c overlap $ov: NTT/s %.0f  pipelined %.4f  pass1 %.4f ms  pass2 %.4f ms (per 1024 transforms)' % (j['value'], r['pipelined_ms_per_batch'], r['pass1_ms_per_batch'], r['pass2_ms_per_batch']))
"
done; done
