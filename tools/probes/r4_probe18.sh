export GPU_MAX_HW_QUEUES=16
CHW_SWEEP=256,auto8,auto16,auto32,auto64,512,1024 timeout 500 python tools/probes/r4_kw_classes.py 2>&1 | grep -v "amdgpu.ids\|first pass" | tail -16
