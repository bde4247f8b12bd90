#!/bin/bash
# A/B bench of alternative builds of libcuhe_hip.so (same ABI): tools/ab_bench.sh lib1.so lib2.so ...
for lib in "$@"; do
  echo "== $lib"
  CUHE_HIP_LIB=$lib python bench.py --steps 10 --warmup 3 --no-mulrelin --no-cpu 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        j = json.loads(line); r = j['roofline']
        print('  NTT/s %.0f  frac %.4f  pipelined %.4f ms  pass1 %.4f ms  pass2 %.4f ms' % (j['value'], r['frac'], r['pipelined_ms_per_batch'], r['pass1_ms_per_batch'], r['pass2_ms_per_batch']))
"
done
