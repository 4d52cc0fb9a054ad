# round 4, second GPU pass (instrumented): where the keyword leg's HOST CPU goes (MSI_SEARCH_CPU_PROFILE), what a wide
# workgroup costs (MSI_VM_PROFILE), where a typo query's time goes inside its workgroup (MSI_DICT_PROFILE, both matchers)
set -x
mkdir -p gpurun_out
export GPU_MAX_HW_QUEUES=16 RB_DETAILED=1 RB_DISTINCT_QUERIES=3072
S='s/"config.*"queries_per_s"/"qps"/'
T='s/"launches_per_query.*"vm"/"vm"/'
U='s/"compact_space.*//'
echo "== host CPU profile: 64, 128, 160 callers (chunks of 256 words, no set cache)"
MSI_SEARCH_CPU_PROFILE=1 timeout 400 tools/bin/ranked_bench 10000000 200000 3 64 64 128 160 > gpurun_out/r4_cpu_profile.jsonl 2> gpurun_out/r4_cpu_profile.err
sed "$S" gpurun_out/r4_cpu_profile.jsonl | sed "$T" | sed "$U" | cut -c1-400
grep 'host CPU' gpurun_out/r4_cpu_profile.err
echo "== opcode + wide profile, 16 callers"
MSI_VM_PROFILE=1 timeout 300 tools/bin/ranked_bench 10000000 200000 3 48 16 2>&1 | grep -o 'msi_vm profile.*' | cut -c1-1200
echo "== C3 bit-parallel, kernel phase profile"
MSI_DICT_PROFILE=1 timeout 300 python bench.py --config c3 --steps 10 --warmup 2 --no-pmc --no-cpu-baseline > gpurun_out/r4_c3_bits2.json 2> gpurun_out/r4_c3_bits2.err; cut -c1-330 gpurun_out/r4_c3_bits2.json; grep 'msi_dict profile' gpurun_out/r4_c3_bits2.err
echo "== C3 banded, kernel phase profile"
MSI_DICT_PROFILE=1 MSI_DICT_MATCHER=banded timeout 300 python bench.py --config c3 --steps 10 --warmup 2 --no-pmc --no-cpu-baseline > gpurun_out/r4_c3_banded2.json 2> gpurun_out/r4_c3_banded2.err; cut -c1-330 gpurun_out/r4_c3_banded2.json; grep 'msi_dict profile' gpurun_out/r4_c3_banded2.err
