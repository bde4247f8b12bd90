#!/bin/bash
# how the pass-1 -> pass-2 slab size (transforms per launch pair) changes each pass's time
for c in ${@:-256 512 1024 2048 4096}; do
  python bench.py --steps 10 --warmup 3 --no-mulrelin --no-cpu --chunk $c 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        j = json.loads(line); r = j['roofline']
        print('chunk $c: NTT/s %.0f  pipelined %.4f  pass1 %.4f ms  pass2 %.4f ms (per 8192 transforms)' % (j['value'], r['pipelined_ms_per_batch'], r['pass1_ms_per_batch'], r['pass2_ms_per_batch']))
"
done
