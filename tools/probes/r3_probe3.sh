export RB_DETAILED=1 RB_DISTINCT_QUERIES=3072
S='s/"config.*"queries_per_s"/"qps"/'
python -m pytest -x -q -m gpu tests/test_configs_gpu.py -k c4_keyword tests/test_zz_vm_gpu.py tests/test_dict_gpu.py 2>&1 | tail -3
echo "== threads 1 16 64 128"; timeout 300 tools/bin/ranked_bench 10000000 200000 3 48 1 16 64 128 2>&1 | sed "$S" | cut -c1-330
echo "== batch wait 50, 64 / 128"; MSI_VM_BATCH_WAIT_US=50 timeout 300 tools/bin/ranked_bench 10000000 200000 3 48 64 128 2>&1 | sed "$S" | cut -c1-200
echo "== vm profile, 64 threads"; MSI_VM_PROFILE=1 timeout 300 tools/bin/ranked_bench 10000000 200000 3 48 64 2>&1 | grep -i "msi_vm profile" | cut -c1-700
