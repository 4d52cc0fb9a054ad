#!/bin/bash
mkdir -p gpurun_out
echo "== VM=0"; MSI_SEARCH_VM=0 timeout 600 python -u -m pytest tests/test_zzz_distinct_gpu.py -m gpu -q -x -s --tb=short -p no:cacheprovider -p no:faulthandler -k "distinct_matches_the_oracle" 2>&1 | tail -4 | cut -c1-300
echo "== default + backtrace"; MSI_DEBUG_ABORT=1 timeout 600 python -u -m pytest tests/test_zzz_distinct_gpu.py -m gpu -q -x -s --tb=short -p no:cacheprovider -p no:faulthandler -k "distinct_matches_the_oracle" 2>&1 | tail -45 | cut -c1-260
