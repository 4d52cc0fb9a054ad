export GPU_MAX_HW_QUEUES=16
REPS=4 CHW_SWEEP=256,1024,auto16,512 timeout 500 python tools/probes/r4_kw_classes.py 2>&1 | grep -v "amdgpu.ids\|first pass" | tail -18
