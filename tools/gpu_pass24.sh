#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_search_gpu.py tests/test_zz_vm_gpu.py tests/test_zz_levels_per_wait_gpu.py tests/test_zzz_distinct_gpu.py tests/test_zz_order_keys_gpu.py tests/test_zzz_geo_gpu.py tests/test_rank_gpu.py tests/test_bits_gpu.py -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/p24_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/p24_tests.log
echo "== detailed"; RB_DETAILED=1 timeout 600 tools/bin/ranked_bench 10000000 200000 3 32 1 16 64 2>/dev/null | sed 's/"config.*"queries_per_s"/"qps"/' | cut -c1-160
echo "== plain"; timeout 600 tools/bin/ranked_bench 10000000 200000 3 32 1 16 64 2>/dev/null | sed 's/"config.*"queries_per_s"/"qps"/' | cut -c1-160
echo "== detailed profile"; RB_DETAILED=1 MSI_VM_PROFILE=1 timeout 600 tools/bin/ranked_bench 10000000 200000 3 16 16 2>&1 >/dev/null | grep "msi_vm profile"
