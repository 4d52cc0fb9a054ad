#!/bin/bash
# one-off: runtime knobs vs keyword-search throughput at 10 M documents, 16 and 48 threads
mkdir -p gpurun_out
run() { echo "== $1"; env $1 timeout 300 tools/bin/ranked_bench 10000000 200000 3 16 16 48 2>/dev/null | sed 's/"config.*"queries_per_s"/"qps"/' | cut -c1-60; }
run "X=1"
run "GPU_MAX_HW_QUEUES=2"
run "GPU_MAX_HW_QUEUES=4"
run "GPU_MAX_HW_QUEUES=16"
run "AMD_OPT_FLUSH=0"
run "ROC_SYSTEM_SCOPE_SIGNAL=0"
run "AMD_DIRECT_DISPATCH=0"
run "ROC_ACTIVE_WAIT_TIMEOUT=0"
run "GPU_FLUSH_ON_EXECUTION=1"
