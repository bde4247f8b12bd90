#!/bin/bash
# tools/bench_matrix.sh "steps warmup [extra args]" ... : one summary line per configuration
for cfg in "$@"; do
  set -- $cfg; s=$1; w=$2; shift 2
  t0=$(date +%s.%N)
  python bench.py --steps $s --warmup $w "$@" 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        j = json.loads(line); r = j['roofline']
        mr = j.get('mul_relin') or {}; mf = j.get('mul_full') or {}
        print('steps $s warmup $w $*:', j['value'], 'frac', r['frac'], 'ms/step', j['ms_per_step'], 'mulrelin', mr.get('ms'), 'mulfull', mf.get('ms'))
"
  echo "   wall $(echo "$(date +%s.%N) - $t0" | bc) s"
done
