#!/bin/bash
# first against second PRINCE block of a process on the library's default gates (unchanged client pattern, one thread): kernel trace cut to
# the timed parts (CUHE_TRACE_MARK=1 + rocpd_summary.py --between), kernels, idle gaps by size and by the kernel that follows
export TMPDIR=/tmp
R=$PWD
make -C cuhe_amd/cxx -s test > /dev/null 2>&1
cd /tmp; rm -rf /tmp/pfb
CUHE_TRACE_MARK=1 timeout 300 rocprofv3 --kernel-trace -d /tmp/pfb -o s -- $R/cuhe_amd/lib/test_prince_flow --threads 1 --default --no-round-checks --repeat 3 2>&1 | grep -E "Prince Encryption|PASSED|recorded"
python $R/tools/rocpd_summary.py --between k_probe_valu /tmp/pfb/s_results.db 2>&1 | grep -v "^void\|^cuhe::\|^__amd" | cut -c1-200
