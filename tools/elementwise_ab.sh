#!/bin/bash
# A/B of the elementwise CRT-domain kernels with 16-byte accesses (CUHE_ELEMENTWISE_VEC=1, the default) against one element
# per thread (=0): PRINCE on arrays (BASELINE config 5; 5 runs each, alternating) and the kernels' times in a kernel trace.
export TMPDIR=/tmp
R=$PWD; out=$R/gpurun_out/elem_ab; mkdir -p $out
cd /tmp
for rep in 1 2 3 4 5; do for on in 1 0; do
  CUHE_ELEMENTWISE_VEC=$on $R/cuhe_amd/lib/test_prince_arrays_cxx --no-round-checks --async --json --devices 1 2>&1 | grep "^{" | sed "s/^/vec=$on /"
done; done 2>&1 | cut -c1-110 | tee $out/elem_ab.txt
for on in 1 0; do
  rm -rf /tmp/ea; CUHE_ELEMENTWISE_VEC=$on timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/ea -o s -- $R/cuhe_amd/lib/test_prince_arrays_cxx --no-round-checks --async --json --devices 1 > /dev/null 2>&1
  python $R/tools/rocpd_summary.py /tmp/ea/s_results.db 2>&1 | grep -E "k_modswitch|k_crt_combine" | cut -c1-60,100-200 | sed "s/^/vec=$on: /"
done 2>&1 | tee -a $out/elem_ab.txt
