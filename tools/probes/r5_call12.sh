# Round 5: the whole GPU tier + the device fuzz legs (the forced bucket-space mode against the REAL kernels)
set -x
mkdir -p gpurun_out
(time timeout 1700 python -m pytest tests -m gpu -x -q) 2>&1 | tail -12
MSI_SEARCH_LATE_COMPACT=2 timeout 260 python tools/fuzz_ranked_hostlogic.py 700000 200 --device 2>&1 | grep -v amdgpu.ids | tail -4
MSI_SEARCH_LATE_COMPACT=2 FUZZ_SPREAD=70000 timeout 160 python tools/fuzz_ranked_hostlogic.py 710000 100 --device 2>&1 | grep -v amdgpu.ids | tail -4
FUZZ_SPREAD=70000 timeout 120 python tools/fuzz_ranked_hostlogic.py 720000 60 --device 2>&1 | grep -v amdgpu.ids | tail -4
