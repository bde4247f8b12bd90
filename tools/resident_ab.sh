L=cuhe_amd/lib; OUT=gpurun_out/resident_ab.txt; : > $OUT
for rep in 1 2 3; do
for v in "X=default" "CUHE_SCHED_LATENCY=0" "CUHE_CLIENT_STAGING=0"; do
  echo "device-resident, 1 client thread, $v: $(env $v timeout 300 $L/test_prince_flow --threads 1 --default --no-round-checks --repeat 4 2>&1 | grep -E "Prince Enc" | sed -E 's/Prince Encryption: ([0-9.]+) s.*/\1/' | tr '\n' ' ')" >> $OUT
done; done
CUHE_SCHED_STATS=1 $L/test_prince_flow --threads 1 --sched --no-round-checks --repeat 2 2>&1 | grep -E "Prince Enc|allocator|task blocks|batches|scheduler:" >> $OUT
CUHE_CLIENT_STAGING=0 CUHE_SCHED_STATS=1 $L/test_prince_flow --threads 1 --sched --no-round-checks --repeat 2 2>&1 | grep -E "Prince Enc|allocator|task blocks|batches|scheduler:" >> $OUT
cat $OUT
