#!/bin/bash
# A/B of alternative builds of libcuhe_hip.so on the metric, alternating on one box: tools/lib_ab.sh libA.so libB.so ...
R=$PWD
for rep in 1 2; do for lib in "$@"; do
CUHE_HIP_LIB=$R/cuhe_amd/lib/$lib timeout 100 python bench.py --steps 30 --warmup 5 --no-mulrelin --no-cpu --no-prince 2>/dev/null | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.read()); r = j['roofline']
print('$lib: NTT/s', j['value'], 'ms/step', j['ms_per_step'], 'pair', r['pipelined_ms_per_batch'], 'p1', r['pass1_ms_per_batch'], 'p2', r['pass2_ms_per_batch'])"
done; done
