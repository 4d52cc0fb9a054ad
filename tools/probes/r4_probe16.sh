export GPU_MAX_HW_QUEUES=16
CLASSES=4 REPS=30 MSI_VM_PROFILE=1 timeout 500 python tools/probes/r4_kw_classes.py 2>&1 | grep -v "amdgpu.ids\|first pass" | tail -4
CLASSES=0 REPS=30 MSI_VM_PROFILE=1 timeout 500 python tools/probes/r4_kw_classes.py 2>&1 | grep -v "amdgpu.ids\|first pass" | tail -4
