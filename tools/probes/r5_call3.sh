# Round 5, GPU call 3: the int8 candidate sweep on hardware (tests, C2 / C4 vector leg), the keyword leg's host CPU on a fresh
# query stream, the round-4 collapse (sorted order, lists of up to 160 chunks fused, round 4's budget).
set -x
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_vs_gpu.py tests/test_zzz_vs_update_gpu.py -m gpu -q -x 2>&1 | tail -5
timeout 300 python bench.py --config c2 --no-pmc 2>/dev/null | tail -1 | cut -c1-1500
timeout 400 python bench.py --config c4 --no-rank --no-typo --no-pmc --steps 5 --warmup 2 2>/dev/null | tail -1 | cut -c1-1800
MSI_SEARCH_CPU_PROFILE=1 timeout 500 python tools/kw_leg.py --queries 3072 --fresh 4608 2>&1 | grep -v amdgpu.ids | tail -2
MSI_VM_FUSED_WGS_PCT=400 MSI_VM_PROFILE=1 FUSE_SWEEP=160 timeout 300 python tools/probes/r4_kw_classes.py 2>&1 | grep -v "amdgpu.ids\|first pass" | tail -6
