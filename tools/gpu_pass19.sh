#!/bin/bash
timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pmc 2>&1 | grep -v amdgpu.ids | tail -6 | cut -c1-600
echo "== RB_DETAILED 3072 distinct"; RB_DISTINCT_QUERIES=3072 RB_DETAILED=1 timeout 600 tools/bin/ranked_bench 10000000 200000 3 48 64 2>&1 | sed 's/"config.*"queries_per_s"/"qps"/' | cut -c1-300 | tail -4
