# Round 6, second device call: the changed kernels and host paths on the device (vector selection in slices / prefetching
# rescoring, lock-free list submission, postings staged at index-open), then the keyword leg with and without staging and
# the headline
set -x
mkdir -p gpurun_out
timeout 1500 python -m pytest -x -q -m gpu tests/test_vs_gpu.py tests/test_zz_vm_gpu.py tests/test_search_gpu.py "tests/test_configs_gpu.py::test_postings_staged_at_index_open_on_the_coherent_corpus" "tests/test_configs_gpu.py::test_c4_10m_x_768_top20" "tests/test_configs_gpu.py::test_c2_1m_x_384_top20" "tests/test_configs_gpu.py::test_c5_shard_bf16_filtered_k1000" 2>&1 | tail -15 > gpurun_out/r6_second_tests.log
cat gpurun_out/r6_second_tests.log
for cfg in "1 " "1 --cold" "0 " "0 --cold"; do
  set -- $cfg
  MSI_SEARCH_CPU_PROFILE=1 timeout 900 python tools/kw_leg.py --callers 256 --queries 3072 --fresh 4608 --stage $1 $2 2>&1 | grep -v amdgpu.ids | tail -1
done > gpurun_out/r6_kw_stage.log 2>&1
cat gpurun_out/r6_kw_stage.log
timeout 700 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-also --no-pmc 2>/dev/null | tail -1 > gpurun_out/r6_bench_second.json
cat gpurun_out/r6_bench_second.json | cut -c1-2500
