# Round 6, tenth device call: chunks per workgroup of the full-space lists (MSI_VM_SPAN 1 / 2 / 4 / 8), each its own process on
# one box; the workgroup profile at the default; the device tests of the lists
set -x
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
timeout 900 python -m pytest -x -q -m gpu tests/test_zz_vm_gpu.py tests/test_search_gpu.py tests/test_zz_levels_per_wait_gpu.py tests/test_bits_gpu.py "tests/test_configs_gpu.py::test_c4_keyword_leg_on_the_coherent_corpus" "tests/test_configs_gpu.py::test_postings_staged_at_index_open_on_the_coherent_corpus" "tests/test_configs_gpu.py::test_phrases_on_the_coherent_corpus" 2>&1 | grep -a "passed\|failed\|error" | tail -3 | tee gpurun_out/r6_tenth_tests.log
for sp in 1 2 4 8; do
  MSI_VM_SPAN=$sp MSI_SEARCH_CPU_PROFILE=1 timeout 900 python tools/kw_leg.py --callers 256 --queries 3072 --fresh 4608 2>&1 | grep -a -v amdgpu.ids | tail -1 | sed "s/^/span=$sp /"
done | tee gpurun_out/r6_span.log | cut -c1-700
MSI_VM_PROFILE=1 timeout 900 python tools/kw_leg.py --callers 256 --queries 3072 --fresh 4608 2>&1 | grep -a "msi_vm profile" | tee gpurun_out/r6_vm_profile_span.log | cut -c1-900
