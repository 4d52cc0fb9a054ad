export RB_DETAILED=1 RB_DISTINCT_QUERIES=3072
S='s/"config.*"queries_per_s"/"qps"/'
T='s/"launches_per_query.*"vm"/"vm"/'
python -m pytest -x -q -m gpu tests/test_configs_gpu.py -k c4_keyword tests/test_zz_vm_gpu.py 2>&1 | tail -2
echo "== 1 / 64 / 128 threads, 96 queries each"; timeout 300 tools/bin/ranked_bench 10000000 200000 3 96 1 64 128 2>&1 | sed "$S" | sed "$T" | cut -c1-520
echo "== vm profile, 64 threads"; MSI_VM_PROFILE=1 timeout 300 tools/bin/ranked_bench 10000000 200000 3 96 64 2>&1 | grep -i "msi_vm profile" | cut -c1-700
