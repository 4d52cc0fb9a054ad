export RB_DETAILED=1 RB_DISTINCT_QUERIES=3072
mkdir -p gpurun_out
S='s/"config.*"queries_per_s"/"qps"/'
T='s/"launches_per_query.*"cpu"/"cpu"/'
echo "== default"; timeout 300 tools/bin/ranked_bench 10000000 200000 3 48 64 2>&1 | sed "$S" | sed "$T" | cut -c1-400
echo "== MSI_PCACHE_KNOWN=0"; MSI_PCACHE_KNOWN=0 timeout 300 tools/bin/ranked_bench 10000000 200000 3 48 64 2>&1 | sed "$S" | sed "$T" | cut -c1-400
echo "== compaction off"; MSI_SEARCH_COMPACT=0 timeout 300 tools/bin/ranked_bench 10000000 200000 3 48 64 2>&1 | sed "$S" | sed "$T" | cut -c1-400
echo "== default again, 96 queries per thread"; timeout 300 tools/bin/ranked_bench 10000000 200000 3 96 64 2>&1 | sed "$S" | sed "$T" | cut -c1-400
