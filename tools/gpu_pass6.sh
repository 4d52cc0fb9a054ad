#!/bin/bash
# one-off: after the fence change and the one-sweep BQ search: device tests of the touched areas, keyword search
# throughput (2 M and 10 M documents), BQ 10 M x 768
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_bits_gpu.py tests/test_search_gpu.py tests/test_zz_vm_gpu.py tests/test_zz_bq_gpu.py tests/test_zz_levels_per_wait_gpu.py tests/test_zz_order_keys_gpu.py tests/test_zzz_distinct_gpu.py tests/test_zzz_geo_gpu.py tests/test_zzz_filter_gpu.py -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/p6_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/p6_tests.log
timeout 300 tools/bin/ranked_bench 2000000 200000 3 32 1 16 64 > gpurun_out/p6_ranked_2m.jsonl 2> gpurun_out/p6_ranked_2m.err; echo rc=$?
timeout 400 tools/bin/ranked_bench 10000000 200000 3 24 1 16 64 > gpurun_out/p6_ranked_10m.jsonl 2> gpurun_out/p6_ranked_10m.err; echo rc=$?
sed 's/"config.*"queries_per_s"/"qps"/' gpurun_out/p6_ranked_2m.jsonl gpurun_out/p6_ranked_10m.jsonl | cut -c1-330
timeout 600 python tools/bench_configs.py bq > gpurun_out/p6_bq_10m.jsonl 2> gpurun_out/p6_bq.err; echo rc=$?; cat gpurun_out/p6_bq_10m.jsonl
