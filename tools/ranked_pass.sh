# the ranked keyword search (all rules) on the device: tests, then serving throughput by caller threads
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_search_gpu.py tests/test_zz_levels_per_wait_gpu.py tests/test_zz_order_keys_gpu.py tests/test_zzz_distinct_gpu.py tests/test_zzz_geo_gpu.py tests/test_rank_gpu.py -m gpu -x -q 2>&1 | tail -4
timeout 300 tools/bin/ranked_bench 2000000 200000 3 48 1 8 16 32 64 128 > gpurun_out/ranked_2m_vm.jsonl 2> gpurun_out/ranked_2m_vm.err; echo rc=$?; cut -c1-420 gpurun_out/ranked_2m_vm.jsonl; tail -2 gpurun_out/ranked_2m_vm.err
MSI_SEARCH_VM=0 timeout 300 tools/bin/ranked_bench 2000000 200000 3 48 1 16 > gpurun_out/ranked_2m_direct.jsonl 2>&1; cut -c1-420 gpurun_out/ranked_2m_direct.jsonl
timeout 300 tools/bin/ranked_bench 10000000 200000 3 32 1 16 64 > gpurun_out/ranked_10m_vm.jsonl 2> gpurun_out/ranked_10m_vm.err; echo rc=$?; cut -c1-420 gpurun_out/ranked_10m_vm.jsonl; tail -2 gpurun_out/ranked_10m_vm.err
