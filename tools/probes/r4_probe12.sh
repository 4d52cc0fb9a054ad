export GPU_MAX_HW_QUEUES=16
echo "== class 2 alone, 160 callers, 64 queries"; CLASSES=2 MAX_PER_CLASS=64 timeout 200 python tools/probes/r4_kw_classes.py 2>&1 | grep -v amdgpu.ids | tail -5
echo "== same, MSI_VM_FUSE=0"; MSI_VM_FUSE=0 CLASSES=2 MAX_PER_CLASS=64 timeout 200 python tools/probes/r4_kw_classes.py 2>&1 | grep -v amdgpu.ids | tail -5
echo "== same, 16 callers"; CALLERS=16 CLASSES=2 MAX_PER_CLASS=64 timeout 200 python tools/probes/r4_kw_classes.py 2>&1 | grep -v amdgpu.ids | tail -5
