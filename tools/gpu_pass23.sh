#!/bin/bash
echo "== detailed"; RB_DETAILED=1 MSI_VM_PROFILE=1 timeout 600 tools/bin/ranked_bench 10000000 200000 3 16 16 2>&1 >/dev/null | grep "msi_vm profile"
echo "== plain"; MSI_VM_PROFILE=1 timeout 600 tools/bin/ranked_bench 10000000 200000 3 16 16 2>&1 >/dev/null | grep "msi_vm profile"
