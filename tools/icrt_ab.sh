#!/bin/bash
# A/B of the inverse CRT forms (cuhe_hip_set_icrt_mfma / CUHE_ICRT_MFMA): kernel time inside the batched multiply +
# relinearise call of BASELINE config 4 on both rings, and the call's per-ciphertext time.  Run on the GPU box from the repo root.
export TMPDIR=/tmp
R=$PWD; out=$R/gpurun_out/icrt_ab; mkdir -p $out
cd /tmp
for ring in 2^15 2^16; do
  for on in 1 0; do
    for rep in 1 2; do CUHE_ICRT_MFMA=$on python $R/tools/trace_batched.py 32 20 $ring 2>&1 | grep "per ciphertext" | sed "s/^/icrt_mfma=$on /"; done
    rm -rf /tmp/ia; CUHE_ICRT_MFMA=$on timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/ia -o s -- python $R/tools/trace_batched.py 32 10 $ring > /dev/null 2>&1
    python $R/tools/rocpd_summary.py /tmp/ia/s_results.db 2>&1 | grep -E "k_icrt" | cut -c1-60,84-200 | sed "s/^/icrt_mfma=$on ring $ring: /"
  done
done 2>&1 | tee $out/icrt_ab.txt
