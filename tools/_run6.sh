cd $GRAFT_REPO_ROOT
L=$PWD/cuhe_amd/lib
export CUHE_SCHED_STATS=1
{
timeout 900 $L/test_prince_flow --threads 1 --sched 3 --no-round-checks --profile 2>&1 | grep -v "^DHS\|^plain\|^encrypted"
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/pq -o s -- $L/test_prince_flow --threads 1 --sched 3 --no-round-checks 2>&1 | grep "Prince Enc"
python3 $GRAFT_REPO_ROOT/tools/rocpd_summary.py /tmp/pq/s_results.db 2>&1 | head -32 | cut -c1-84,112-160
} > $GRAFT_REPO_ROOT/gpurun_out/run6.txt 2>&1
