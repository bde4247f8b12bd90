#!/bin/bash
# why some PRINCE blocks of a process take 0.066-0.075 s and others 0.058-0.060 s on the same gates: kernel trace of REPEAT blocks cut to the timed parts
# (CUHE_TRACE_MARK=1 + rocpd_summary.py --between): per block the wall time, the time some kernel was running, the summed kernel time and the idle gaps
export TMPDIR=/tmp
R=$PWD
make -C cuhe_amd/cxx -s test > /dev/null 2>&1
cd /tmp
for run in 1 2 3; do
  rm -rf /tmp/pfb
  CUHE_TRACE_MARK=1 timeout 300 rocprofv3 --kernel-trace -d /tmp/pfb -o s -- $R/cuhe_amd/lib/test_prince_flow --threads 1 --default --no-round-checks --repeat ${1:-6} 2>&1 | grep -E "Prince Encryption" | sed -E 's/ on 1 device.*//' | tr '\n' ' '; echo
  python $R/tools/rocpd_summary.py --between k_probe_valu /tmp/pfb/s_results.db 2>&1 | grep -E "stretch|idle gaps|ntt_onewg<14, 0, 0, true>|k_relin_mac_mfma" | cut -c1-190
done
