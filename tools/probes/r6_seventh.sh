# Round 6, seventh device call: row tiles with a row's pieces of a block in one 64-byte sector (MSI_TILE_PIECE) — device tests,
# the vector leg's traces at C4's and C2's shapes (the f32 sweep reads the permuted block: its time is the thing to watch)
set -x
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
timeout 1500 python -m pytest -x -q -m gpu tests/test_vs_gpu.py tests/test_zz_i8_proof_gpu.py tests/test_zzz_vs_update_gpu.py tests/test_rank_gpu.py tests/test_zz_group_gpu.py "tests/test_configs_gpu.py::test_c4_10m_x_768_top20" "tests/test_configs_gpu.py::test_c2_1m_x_384_top20" "tests/test_configs_gpu.py::test_c2_with_10pct_filter" "tests/test_configs_gpu.py::test_c5_shard_bf16_filtered_k1000" 2>&1 | tail -4 > gpurun_out/r6_seventh_tests.log
cat gpurun_out/r6_seventh_tests.log
cd /tmp && export TMPDIR=/tmp
for shape in "c4 10000000 768 768" "c2 1000000 384 256"; do
  set -- $shape
  rm -rf /tmp/tr_$1
  N_ROWS=$2 DIM=$3 Q=$4 VARIANTS=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr_$1 -o tr -- python $R/tools/probes/r5_i8_variants.py > /tmp/tr_$1.log 2>&1
  F=$(find /tmp/tr_$1 -name "*kernel_stats.csv" | head -1)
  [ -n "$F" ] && cp $F $R/gpurun_out/r6_vector_leg_$1_kernel_stats_d.csv && head -14 $F | cut -c1-200
  grep -a "variant 0\|f32 level" /tmp/tr_$1.log | tee -a $R/gpurun_out/r6_vector_leg_lines_d.log
done
cd $R
timeout 600 python bench.py --config c5 --no-pmc 2>/dev/null | tail -1 | cut -c1-1500 | tee gpurun_out/r6_bench_c5_d.json
timeout 700 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-also --no-pmc --kw-features 0 2>/dev/null | tail -1 | cut -c1-3000 | tee gpurun_out/r6_bench_c4_d.json
