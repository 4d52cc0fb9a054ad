export GPU_MAX_HW_QUEUES=16
timeout 500 python tools/probes/r4_kw_classes.py 2>&1 | grep -v "amdgpu.ids\|first pass" | tail -12
