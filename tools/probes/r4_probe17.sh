export GPU_MAX_HW_QUEUES=16
echo "== late compaction on (default)"; timeout 500 python tools/probes/r4_kw_classes.py 2>&1 | grep -v "amdgpu.ids\|first pass" | tail -9
echo "== late compaction off"; MSI_SEARCH_LATE_COMPACT=0 timeout 500 python tools/probes/r4_kw_classes.py 2>&1 | grep -v "amdgpu.ids\|first pass" | tail -9
