export RB_DETAILED=1 RB_DISTINCT_QUERIES=3072 GPU_MAX_HW_QUEUES=16
S='s/"config.*"queries_per_s"/"qps"/'
T='s/"launches_per_query.*"vm"/"vm"/'
U='s/"compact_space.*//'
for W in 200 400; do for CAP in 32 64; do
echo "== wait $W cap $CAP"; MSI_VM_BATCH_WAIT_US=$W MSI_VM_BATCH_CAP=$CAP timeout 300 tools/bin/ranked_bench 10000000 200000 3 96 128 2>&1 | sed "$S" | sed "$T" | sed "$U" | cut -c1-330
done; done
echo "== div 1 (wait for all in flight), cap 64, wait 300"; MSI_VM_BATCH_DIV=1 MSI_VM_BATCH_WAIT_US=300 MSI_VM_BATCH_CAP=64 timeout 300 tools/bin/ranked_bench 10000000 200000 3 96 128 2>&1 | sed "$S" | sed "$T" | sed "$U" | cut -c1-330
echo "== algorithmic bytes, 1 thread"; timeout 300 tools/bin/ranked_bench 10000000 200000 3 48 1 2>&1 | grep -o '"compact_space.*' 
echo "== algorithmic bytes, 1 thread, no compaction"; MSI_SEARCH_COMPACT=0 timeout 300 tools/bin/ranked_bench 10000000 200000 3 48 1 2>&1 | grep -o '"compact_space.*' 
