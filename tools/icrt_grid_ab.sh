#!/bin/bash
# A/B of the launch grid of k_icrt_mfma (resident workgroups walking the tiles with a grid stride, x CUHE_ICRT_GRID_MULT): kernel time inside
# the batched multiply + relinearise call of BASELINE config 4 (32 ciphertexts per call), both rings.  Output: gpurun_out/icrt_grid_ab.txt
export TMPDIR=/tmp
R=$PWD; OUT=$R/gpurun_out/icrt_grid_ab.txt; : > $OUT
cd /tmp
for ring in 2^15 2^16; do for mult in 1 2 4 16; do
  rm -rf /tmp/ig; CUHE_ICRT_GRID_MULT=$mult timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/ig -o s -- python $R/tools/trace_batched.py 32 10 $ring > /tmp/ig.txt 2>&1
  echo "ring $ring grid x$mult: $(grep 'per ciphertext' /tmp/ig.txt)  $(python $R/tools/rocpd_summary.py /tmp/ig/s_results.db 2>&1 | grep -E 'k_icrt' | cut -c1-40,112-175)" >> $OUT
done; done
cat $OUT
