set -x
export GPU_MAX_HW_QUEUES=16
MSI_VM_PROFILE=1 timeout 900 python tools/probes/r4_kw_classes.py 2>&1 | tail -30
