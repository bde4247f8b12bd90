#!/bin/bash
# A/B of the first-operand fragments of k_relin_mac_mfma: 4 x 4 byte transposes with v_perm_b32 (default since round 6) against the
# shift / mask / or form (libcuhe_hip_packdigit.so = tools/build_variant.py ... -DCUHE_MAC_PACK_DIGIT), alternating on one box:
# the batched multiply + relinearise of bench.py, 32 ciphertexts per call, both config-4 rings.  Output: gpurun_out/mac_perm_ab.txt
R=$PWD; OUT=gpurun_out/mac_perm_ab.txt; mkdir -p gpurun_out; : > $OUT
for rep in 1 2 3; do for lib in libcuhe_hip.so libcuhe_hip_packdigit.so; do for ring in 2^15 2^16; do
  echo "$lib: $(CUHE_HIP_LIB=$R/cuhe_amd/lib/$lib timeout 200 python tools/trace_batched.py 32 20 $ring 2>/dev/null | tail -1)" >> $OUT
done; done; done
cat $OUT
