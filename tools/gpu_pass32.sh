#!/bin/bash
run() { echo "== $1"; env $1 RB_DETAILED=1 timeout 600 tools/bin/ranked_bench 10000000 200000 3 32 16 64 2>/dev/null | sed 's/"config.*"queries_per_s"/"qps"/' | cut -c1-60; }
run "MSI_VM_BATCH_WAIT_US=120"
run "MSI_VM_BATCH_WAIT_US=200"
run "MSI_VM_BATCH_WAIT_US=120 MSI_VM_BATCH_DIV=2 MSI_VM_BATCH_CAP=32"
run "MSI_VM_BATCH_WAIT_US=200 MSI_VM_BATCH_DIV=2 MSI_VM_BATCH_CAP=32"
run "MSI_VM_BATCH_WAIT_US=80 MSI_VM_BATCH_DIV=2 MSI_VM_BATCH_CAP=32"
