#!/bin/bash
mkdir -p gpurun_out
MSI_DEBUG_ABORT=1 timeout 900 python -m pytest tests/test_zzz_distinct_gpu.py -m gpu -q -x --tb=short -p no:cacheprovider -k "distinct_matches_the_oracle" > gpurun_out/p26_a.log 2>&1; echo "rc=$?"; grep -n "msi\] fatal" -A40 gpurun_out/p26_a.log | cut -c1-200 | head -60
