#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_zzz_distinct_gpu.py -m gpu -q -x --tb=short -p no:cacheprovider -p no:faulthandler --capture=sys > gpurun_out/p28_a.log 2>&1; echo "rc=$?"
grep -v "^$" gpurun_out/p28_a.log | grep -iv "amdgpu.ids" | tail -12 | cut -c1-400
