#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
L=cuhe_amd/lib
timeout 300 $L/test_prince_flow --threads 1 --sched --no-round-checks --repeat 3 --profile 2>&1 | python tools/resolve_samples.py | grep -v "^plain\|^DHS\|^encrypted" > gpurun_out/r05_sched_profile2.txt 2>&1
grep -n "Prince Encryption\|^thread\|^samples" gpurun_out/r05_sched_profile2.txt | head -40
awk '/^thread/{c++} c>=9 && c<=12' gpurun_out/r05_sched_profile2.txt | cut -c1-200 | head -120
