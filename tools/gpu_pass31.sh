#!/bin/bash
for w in 0 20 40 80; do echo "== batch wait $w us"; MSI_VM_BATCH_WAIT_US=$w RB_DETAILED=1 timeout 600 tools/bin/ranked_bench 10000000 200000 3 32 1 16 64 2>/dev/null | sed 's/"config.*"queries_per_s"/"qps"/' | cut -c1-60; done
