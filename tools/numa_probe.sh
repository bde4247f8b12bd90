#!/bin/bash
# NUMA placement of the threads that launch: topology of the box, then PRINCE blocks (device-resident client, library default gates, 6 per process)
# with the scheduler's workers left to the kernel's placement, pinned to the device's local CPUs (default), and the whole process under taskset.
L=cuhe_amd/lib; OUT=gpurun_out/numa_probe.txt; : > $OUT
{ echo "nproc $(nproc)"; taskset -p $$; lscpu | grep -iE "numa|socket|model name|^CPU\(s\)"; cat /sys/fs/cgroup/cpu.max 2>/dev/null; cat /sys/fs/cgroup/cpuset.cpus.effective 2>/dev/null
for d in /sys/class/drm/card*/device; do echo "$d vendor $(cat $d/vendor) numa_node $(cat $d/numa_node 2>/dev/null) local_cpulist $(cat $d/local_cpulist 2>/dev/null)"; done
ls /sys/class/kfd/kfd/topology/nodes/ 2>/dev/null | head -20
which numactl taskset; } >> $OUT 2>&1
LOCAL=$(for d in /sys/class/drm/card*/device; do if [ "$(cat $d/vendor)" = "0x1002" ]; then cat $d/local_cpulist; break; fi; done)
echo "local cpulist of the first AMD card: $LOCAL" >> $OUT
run() { echo "$1: $( "${@:2}" $L/test_prince_flow --threads 1 --default --no-round-checks --repeat 6 2>&1 | grep -E "Prince Enc" | sed -E 's/Prince Encryption: ([0-9.]+) s.*/\1/' | tr '\n' ' ')" >> $OUT; }
for rep in 1 2 3 4 5; do
  run "nothing pinned (CUHE_SCHED_PIN=0 CUHE_PIN_CLIENT=0)       " env CUHE_SCHED_PIN=0 CUHE_PIN_CLIENT=0
  run "workers pinned only (CUHE_PIN_CLIENT=0)                   " env CUHE_PIN_CLIENT=0
  run "library default (workers pinned, client local during init)" env X=1
  run "client stays pinned too (CUHE_PIN_CLIENT=1)               " env CUHE_PIN_CLIENT=1
  [ -n "$LOCAL" ] && [ $rep -le 2 ] && run "taskset local, whole process                              " taskset -c $LOCAL
done
cat $OUT
