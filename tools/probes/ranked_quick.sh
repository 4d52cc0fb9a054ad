mkdir -p gpurun_out
timeout 300 tools/bin/ranked_bench 2000000 200000 3 32 1 16 64 96 2>&1 | sed 's/"config.*"queries_per_s"/"qps"/' | cut -c1-1400
echo "10M docs"; timeout 300 tools/bin/ranked_bench 10000000 200000 3 24 1 64 2>&1 | sed 's/"config.*"queries_per_s"/"qps"/' | cut -c1-1400
