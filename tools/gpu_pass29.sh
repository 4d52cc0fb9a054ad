#!/bin/bash
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q -x --tb=short -p no:cacheprovider > gpurun_out/p29_tests.log 2>&1; echo "tests rc=$?"; grep -v "^$" gpurun_out/p29_tests.log | grep -iv "amdgpu.ids\|Librccl\|RCCL version\|HIP version\|ROCm version\|Hostname" | tail -5 | cut -c1-300
