#!/bin/bash
for n in 1 2 4; do echo "== combiners $n"; MSI_VM_COMBINERS=$n RB_DETAILED=1 timeout 600 tools/bin/ranked_bench 10000000 200000 3 32 64 128 2>/dev/null | sed 's/"config.*"queries_per_s"/"qps"/' | cut -c1-60; done
