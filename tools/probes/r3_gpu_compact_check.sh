mkdir -p gpurun_out
(time python -m pytest -x -q -m gpu tests/test_configs_gpu.py::test_c4_keyword_leg tests/test_search_gpu.py tests/test_zz_vm_gpu.py tests/test_zzz_distinct_gpu.py 2>&1 | tail -15) 2>&1
echo "== forced compaction"
(MSI_SEARCH_COMPACT=2 python -m pytest -x -q -m gpu tests/test_search_gpu.py tests/test_zz_vm_gpu.py 2>&1 | tail -5)
echo "== throughput by caller threads (compaction on)"
export RB_DETAILED=1 RB_DISTINCT_QUERIES=3072
timeout 400 tools/bin/ranked_bench 10000000 200000 3 24 1 16 64 128 2>&1 | tee gpurun_out/r3_ranked_10m_compact.jsonl | sed 's/"config.*"queries_per_s"/"qps"/' | cut -c1-1500
echo "== compaction off"
MSI_SEARCH_COMPACT=0 timeout 400 tools/bin/ranked_bench 10000000 200000 3 24 64 2>&1 | sed 's/"config.*"queries_per_s"/"qps"/' | cut -c1-600
