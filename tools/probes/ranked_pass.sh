# the ranked keyword search (all rules) on the device: tests, then serving throughput by caller threads
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_search_gpu.py tests/test_zz_vm_gpu.py tests/test_zz_levels_per_wait_gpu.py tests/test_zz_order_keys_gpu.py tests/test_zzz_distinct_gpu.py tests/test_zzz_geo_gpu.py tests/test_rank_gpu.py tests/test_bits_gpu.py -m gpu -x -q 2>&1 | tail -4
timeout 300 tools/bin/ranked_bench 2000000 200000 3 48 1 8 16 32 64 96 > gpurun_out/ranked_2m_vm.jsonl 2> gpurun_out/ranked_2m_vm.err; echo rc=$?; cut -c1-200 gpurun_out/ranked_2m_vm.jsonl; tail -2 gpurun_out/ranked_2m_vm.err
timeout 300 tools/bin/ranked_bench 2000000 200000 1 48 1 64 > gpurun_out/ranked_2m_vm_1term.jsonl 2>&1; cut -c1-200 gpurun_out/ranked_2m_vm_1term.jsonl
timeout 300 tools/bin/ranked_bench 2000000 200000 5 32 1 64 > gpurun_out/ranked_2m_vm_5terms.jsonl 2>&1; cut -c1-200 gpurun_out/ranked_2m_vm_5terms.jsonl
MSI_SEARCH_VM=0 timeout 300 tools/bin/ranked_bench 2000000 200000 3 48 1 16 > gpurun_out/ranked_2m_direct.jsonl 2>&1; cut -c1-200 gpurun_out/ranked_2m_direct.jsonl
timeout 300 tools/bin/ranked_bench 10000000 200000 3 32 1 16 64 > gpurun_out/ranked_10m_vm.jsonl 2> gpurun_out/ranked_10m_vm.err; echo rc=$?; cut -c1-200 gpurun_out/ranked_10m_vm.jsonl; tail -2 gpurun_out/ranked_10m_vm.err
