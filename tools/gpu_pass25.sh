#!/bin/bash
mkdir -p gpurun_out
AMD_LOG_LEVEL=1 timeout 900 python -X faulthandler -m pytest tests/test_zzz_distinct_gpu.py -m gpu -q -x -v --tb=short -p no:cacheprovider > gpurun_out/p25_a.log 2>&1; echo "rc=$?"; grep -v "^$" gpurun_out/p25_a.log | grep -v "File \"/usr" | tail -25 | cut -c1-300
echo "== summaries off"; MSI_VM_SUMMARY=0 timeout 900 python -m pytest tests/test_zzz_distinct_gpu.py -m gpu -q -x --tb=short -p no:cacheprovider 2>&1 | tail -2
which gdb
