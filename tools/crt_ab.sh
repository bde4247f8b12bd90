#!/bin/bash
# A/B of the CRT kernel forms (CUHE_CRT_ACC64=1: 64-bit sums without a carry word, k_crt<true>; 0: 96-bit sums, k_crt<false>):
# the full multiply raw -> raw of BASELINE config 3 (bench.py's mul_full leg, one at a time and batched by 16), alternating
# processes on one box, and the kernels' times in a kernel trace of the same command.  Run on the GPU box from the repo root.
export TMPDIR=/tmp
R=$PWD; out=$R/gpurun_out/crt_ab; mkdir -p $out
cd /tmp
for on in 1 0 1 0; do
  CUHE_CRT_ACC64=$on python $R/bench.py --steps 5 --warmup 2 --no-cpu --no-prince --one-ring 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); m=d['mul_full']; print('crt_acc64=$on mul_full ms', m['ms'], 'batched ms per multiply', m['batched']['ms_per_multiply'], m['batched']['checked'])"
done 2>&1 | tee $out/crt_ab.txt
for on in 1 0; do
  rm -rf /tmp/ca; CUHE_CRT_ACC64=$on timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/ca -o s -- python $R/bench.py --steps 5 --warmup 2 --no-cpu --no-prince --one-ring > /dev/null 2>&1
  python $R/tools/rocpd_summary.py /tmp/ca/s_results.db 2>&1 | grep -E "k_crt" | cut -c1-60,100-200 | sed "s/^/crt_acc64=$on: /"
done 2>&1 | tee -a $out/crt_ab.txt
