# Round 6, fifth device call: the vector leg with the refined second opinion (parallel f32 cosine, reference arithmetic only
# inside its bound) — device tests, kernel traces at C4's and C2's shapes — and the whole default command
set -x
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
timeout 1500 python -m pytest -x -q -m gpu tests/test_vs_gpu.py tests/test_zz_i8_proof_gpu.py tests/test_zzz_vs_update_gpu.py tests/test_rank_gpu.py "tests/test_configs_gpu.py::test_c4_10m_x_768_top20" "tests/test_configs_gpu.py::test_c2_1m_x_384_top20" "tests/test_configs_gpu.py::test_c2_with_10pct_filter" "tests/test_configs_gpu.py::test_c5_shard_bf16_filtered_k1000" 2>&1 | tail -4 > gpurun_out/r6_fifth_tests.log
cat gpurun_out/r6_fifth_tests.log
cd /tmp && export TMPDIR=/tmp
for shape in "c4 10000000 768 768" "c2 1000000 384 256"; do
  set -- $shape
  rm -rf /tmp/tr_$1
  N_ROWS=$2 DIM=$3 Q=$4 VARIANTS=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr_$1 -o tr -- python $R/tools/probes/r5_i8_variants.py > /tmp/tr_$1.log 2>&1
  F=$(find /tmp/tr_$1 -name "*kernel_stats.csv" | head -1)
  [ -n "$F" ] && cp $F $R/gpurun_out/r6_vector_leg_$1_kernel_stats_c.csv && head -16 $F | cut -c1-200
  grep -a "variant 0\|f32 level" /tmp/tr_$1.log | tee -a $R/gpurun_out/r6_vector_leg_lines_c.log
done
cd $R
( time timeout 900 python bench.py > gpurun_out/r6_bench_default.log 2>gpurun_out/r6_bench_default.err ) 2>&1 | tail -3
tail -1 gpurun_out/r6_bench_default.log | cut -c1-4200
grep -a "phase\|\[bench\]" gpurun_out/r6_bench_default.err | tail -40
