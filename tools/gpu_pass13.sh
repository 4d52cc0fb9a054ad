#!/bin/bash
mkdir -p gpurun_out
echo "== 64 distinct"; timeout 400 tools/bin/ranked_bench 10000000 200000 3 48 64 2>/dev/null | sed 's/"config.*"queries_per_s"/"qps"/' | cut -c1-700
echo "== 3072 distinct"; RB_DISTINCT_QUERIES=3072 timeout 600 tools/bin/ranked_bench 10000000 200000 3 48 64 2>/dev/null | sed 's/"config.*"queries_per_s"/"qps"/' | cut -c1-700
echo "== 3072 distinct, 4 hw queues"; GPU_MAX_HW_QUEUES=4 RB_DISTINCT_QUERIES=3072 timeout 600 tools/bin/ranked_bench 10000000 200000 3 48 64 2>/dev/null | sed 's/"config.*"queries_per_s"/"qps"/' | cut -c1-700
