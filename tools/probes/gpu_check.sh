#!/bin/bash
# quick device check of the working tree: the GPU tier, then smoke()
timeout 1200 python -m pytest tests -m gpu -q -x --tb=short -p no:cacheprovider 2>&1 | grep -v "^$" | grep -iv "amdgpu.ids\|Librccl\|RCCL version\|HIP version\|ROCm version\|Hostname" | tail -3 | cut -c1-200
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-200
