#!/bin/bash
# one-off: device tests of the keyword search after the interpreter change, throughput at 2 M / 10 M, in-kernel profile
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_search_gpu.py tests/test_zz_vm_gpu.py tests/test_zz_levels_per_wait_gpu.py tests/test_zz_order_keys_gpu.py tests/test_zzz_distinct_gpu.py tests/test_zzz_geo_gpu.py -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/p10_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/p10_tests.log
timeout 300 tools/bin/ranked_bench 2000000 200000 3 32 1 16 64 > gpurun_out/p10_ranked_2m.jsonl 2> gpurun_out/p10_ranked_2m.err; echo rc=$?
timeout 400 tools/bin/ranked_bench 10000000 200000 3 24 1 16 64 > gpurun_out/p10_ranked_10m.jsonl 2> gpurun_out/p10_ranked_10m.err; echo rc=$?
sed 's/"config.*"queries_per_s"/"qps"/' gpurun_out/p10_ranked_2m.jsonl gpurun_out/p10_ranked_10m.jsonl | cut -c1-200
MSI_VM_PROFILE=1 timeout 400 tools/bin/ranked_bench 10000000 200000 3 16 16 > gpurun_out/p10_prof10.jsonl 2> gpurun_out/p10_prof10.err; grep "msi_vm profile" gpurun_out/p10_prof10.err
