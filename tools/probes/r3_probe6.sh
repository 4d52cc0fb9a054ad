export RB_DETAILED=1 RB_DISTINCT_QUERIES=3072
S='s/"config.*"queries_per_s"/"qps"/'
T='s/"launches_per_query.*"vm"/"vm"/'
python -m pytest -x -q -m gpu tests/test_configs_gpu.py -k c4_keyword tests/test_zz_vm_gpu.py tests/test_search_gpu.py 2>&1 | tail -2
echo "== fused: 1 / 64 / 128 threads"; timeout 300 tools/bin/ranked_bench 10000000 200000 3 96 1 64 128 2>&1 | sed "$S" | sed "$T" | cut -c1-420
echo "== MSI_VM_FUSE=0: 1 / 64 threads"; MSI_VM_FUSE=0 timeout 300 tools/bin/ranked_bench 10000000 200000 3 96 1 64 2>&1 | sed "$S" | sed "$T" | cut -c1-420
echo "== fused, batch wait 50: 64 / 128"; MSI_VM_BATCH_WAIT_US=50 timeout 300 tools/bin/ranked_bench 10000000 200000 3 96 64 128 2>&1 | sed "$S" | sed "$T" | cut -c1-420
