export GPU_MAX_HW_QUEUES=16
FUSE_SWEEP=0,24,48,160 timeout 600 python tools/probes/r4_kw_classes.py 2>&1 | grep -v amdgpu.ids | tail -20
