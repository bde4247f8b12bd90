#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
L=cuhe_amd/lib
{
for i in 1 2; do
timeout 300 $L/test_prince_flow --threads 1 --sched --no-round-checks --profile 2>&1 | python tools/resolve_samples.py | grep -v "^plain\|^DHS\|^encrypted" | head -150
done
} > gpurun_out/r05_sched_profile.txt 2>&1
cat gpurun_out/r05_sched_profile.txt | cut -c1-230
