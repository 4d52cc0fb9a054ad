#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_search_gpu.py tests/test_zz_vm_gpu.py tests/test_zz_levels_per_wait_gpu.py tests/test_zzz_distinct_gpu.py tests/test_zz_order_keys_gpu.py tests/test_zzz_geo_gpu.py tests/test_rank_gpu.py -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/p18_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/p18_tests.log
echo "== detailed"; RB_DETAILED=1 timeout 600 tools/bin/ranked_bench 10000000 200000 3 32 1 16 64 2>/dev/null | sed 's/"config.*"queries_per_s"/"qps"/' | cut -c1-160
echo "== not detailed"; timeout 600 tools/bin/ranked_bench 10000000 200000 3 32 1 16 64 2>/dev/null | sed 's/"config.*"queries_per_s"/"qps"/' | cut -c1-160
echo "== detailed 2M"; RB_DETAILED=1 timeout 600 tools/bin/ranked_bench 2000000 200000 3 32 1 64 2>/dev/null | sed 's/"config.*"queries_per_s"/"qps"/' | cut -c1-160
