export RB_DETAILED=1 RB_DISTINCT_QUERIES=3072
S='s/"config.*"queries_per_s"/"qps"/'
echo "== threads 1 16 64 128 192"; timeout 300 tools/bin/ranked_bench 10000000 200000 3 48 1 16 64 128 192 2>&1 | sed "$S" | cut -c1-260
echo "== 64 threads, batch wait 0";  MSI_VM_BATCH_WAIT_US=0 timeout 300 tools/bin/ranked_bench 10000000 200000 3 48 64 128 2>&1 | sed "$S" | cut -c1-200
echo "== 64 threads, batch wait 50"; MSI_VM_BATCH_WAIT_US=50 timeout 300 tools/bin/ranked_bench 10000000 200000 3 48 64 128 2>&1 | sed "$S" | cut -c1-200
echo "== 64 threads, 2 combiners"; MSI_VM_COMBINERS=2 timeout 300 tools/bin/ranked_bench 10000000 200000 3 48 64 128 2>&1 | sed "$S" | cut -c1-200
echo "== vm profile, 64 threads"; MSI_VM_PROFILE=1 timeout 300 tools/bin/ranked_bench 10000000 200000 3 48 64 2>&1 | grep -i "msi_vm profile" | cut -c1-700
echo "== vm profile, 1 thread"; MSI_VM_PROFILE=1 timeout 300 tools/bin/ranked_bench 10000000 200000 3 48 1 2>&1 | grep -i "msi_vm profile" | cut -c1-700
