# round 4, first GPU pass: the GPU tier on the new kernels (LDS set cache, wide-phase descriptors in LDS, bit-parallel typo
# matcher), the keyword leg's variants side by side in one process, the interpreter's opcode profile, C3 with both matchers,
# and the default bench line
set -x
mkdir -p gpurun_out
export GPU_MAX_HW_QUEUES=16
t0=$(date +%s)
timeout 900 python -m pytest tests -m gpu -q --maxfail=5 2>&1 | tail -8
echo "tests took $(( $(date +%s) - t0 )) s"
export RB_DETAILED=1 RB_DISTINCT_QUERIES=3072
S='s/"config.*"queries_per_s"/"qps"/'
T='s/"launches_per_query.*"vm"/"vm"/'
t0=$(date +%s)
RB_VARIANTS="MSI_VM_CACHE=0,MSI_VM_COMPACT_CHW=1024;MSI_VM_CACHE=1,MSI_VM_COMPACT_CHW=256;MSI_VM_CACHE=1,MSI_VM_COMPACT_CHW=128;MSI_VM_CACHE=1,MSI_VM_COMPACT_CHW=512;MSI_VM_CACHE=0,MSI_VM_COMPACT_CHW=256;MSI_VM_CACHE=1,MSI_VM_COMPACT_CHW=256;MSI_VM_CACHE=0,MSI_VM_COMPACT_CHW=1024;MSI_VM_CACHE=1,MSI_VM_COMPACT_CHW=256" \
  timeout 600 tools/bin/ranked_bench 10000000 200000 3 64 128 128 128 128 128 1 1 160 > gpurun_out/r4_ranked_variants.jsonl 2> gpurun_out/r4_ranked_variants.err
echo "variants took $(( $(date +%s) - t0 )) s"
grep variant gpurun_out/r4_ranked_variants.err
sed "$S" gpurun_out/r4_ranked_variants.jsonl | sed "$T" | cut -c1-420
tail -3 gpurun_out/r4_ranked_variants.err
echo "== opcode profile, cache on, 16 callers"
MSI_VM_PROFILE=1 timeout 300 tools/bin/ranked_bench 10000000 200000 3 48 16 2>&1 | grep -o 'msi_vm profile.*' | cut -c1-900
echo "== opcode profile, cache off chw 1024, 16 callers"
MSI_VM_PROFILE=1 MSI_VM_CACHE=0 MSI_VM_COMPACT_CHW=1024 timeout 300 tools/bin/ranked_bench 10000000 200000 3 48 16 2>&1 | grep -o 'msi_vm profile.*' | cut -c1-900
echo "== C3 bit-parallel"
timeout 300 python bench.py --config c3 --steps 10 --warmup 2 --no-pmc > gpurun_out/r4_c3_bits.json 2> gpurun_out/r4_c3_bits.err; cut -c1-700 gpurun_out/r4_c3_bits.json
echo "== C3 banded"
MSI_DICT_MATCHER=banded timeout 300 python bench.py --config c3 --steps 10 --warmup 2 --no-pmc > gpurun_out/r4_c3_banded.json 2> gpurun_out/r4_c3_banded.err; cut -c1-700 gpurun_out/r4_c3_banded.json
t0=$(date +%s)
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r4_bench_c4_probe1.json 2> gpurun_out/r4_bench_c4_probe1.err; echo bench rc=$?
echo "bench took $(( $(date +%s) - t0 )) s"
cut -c1-1500 gpurun_out/r4_bench_c4_probe1.json; tail -3 gpurun_out/r4_bench_c4_probe1.err
