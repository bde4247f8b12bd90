#!/bin/bash
# list transforms (rows of separately owned blocks inside the one-workgroup kernels): parity, the scheduled PRINCE modes, timing with and without, NTT bench line
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
R=$PWD; L=cuhe_amd/lib
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "separately_owned or list_block" 2>&1 | tail -8
timeout 900 python -m pytest tests/test_gpu_cxx_api.py -q -x -k "scheduled" 2>&1 | tail -4
for v in 1 0 1 0; do
  echo "== CUHE_SCHED_LISTS=$v"
  CUHE_SCHED_LISTS=$v timeout 300 $L/test_prince_flow --threads 1 --sched --no-round-checks --repeat 4 2>&1 | grep -E "Prince Encryption|FAILED" | awk '{printf "%s ", $3} END {print ""}'
done
timeout 300 $L/test_prince_arrays_cxx --no-round-checks --async --repeat 3 2>&1 | grep -E "Prince Encryption|FAILED"
timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu --no-prince --no-limiter --no-pmc 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('NTT/s', d['value'], 'frac', r['frac'], {k:(v.get('value'), v.get('frac')) for k,v in r.get('other_lengths',{}).items()}); print('mul_relin', d['mul_relin']['ms'], d['mul_relin']['batched']['ms_per_ciphertext'], d['mul_relin_other_ring']['ms'], d['mul_relin_other_ring']['batched']['ms_per_ciphertext'])"
export TMPDIR=/tmp CUHE_TRACE_MARK=1
cd /tmp; rm -rf /tmp/ps
timeout 300 rocprofv3 --kernel-trace -d /tmp/ps -o s -- $R/$L/test_prince_flow --threads 1 --sched --no-round-checks --repeat 2 2>&1 | grep -E "Prince Encryption|FAILED"
python $R/tools/rocpd_summary.py --between k_probe_valu /tmp/ps/s_results.db 2>&1 | cut -c1-100,111-160 > $R/gpurun_out/r05_gaps_sched3.txt
grep -n "^--\|idle gaps\|k_move" $R/gpurun_out/r05_gaps_sched3.txt | cut -c1-180
