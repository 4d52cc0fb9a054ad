# Round 6, thirteenth device call: ROW-GRANULAR filtered sweeps (vs_filter_rows_kernel: compacted allowed rows gathered at
# 64-byte-sector granularity / whole tiles, per region), the int8 copy in row-sector order, the int8 copy for bf16 stores.
# Device tests of the vector tier, the vector leg's traces at C4 / C2 (the int8 sweep now reads the permuted block), C5 with
# the gather, without it (MSI_VS_GATHER_PCT=1600: rounds 1-5's tile-granular sweep) and without the bf16 stores' int8 copy.
set -x
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
timeout 1800 python -m pytest -x -q -m gpu tests/test_vs_gpu.py tests/test_zz_i8_proof_gpu.py tests/test_zzz_vs_update_gpu.py tests/test_zzz_filter_gpu.py tests/test_zz_group_gpu.py "tests/test_configs_gpu.py::test_c4_10m_x_768_top20" "tests/test_configs_gpu.py::test_c2_1m_x_384_top20" "tests/test_configs_gpu.py::test_c2_with_10pct_filter" "tests/test_configs_gpu.py::test_c5_shard_bf16_filtered_k1000" 2>&1 | grep -a "passed\|failed\|error\|Error\|assert" | tail -12 | tee gpurun_out/r6_thirteenth_tests.log
cd /tmp && export TMPDIR=/tmp
for shape in "c4 10000000 768 768" "c2 1000000 384 256"; do
  set -- $shape
  rm -rf /tmp/tr_$1
  N_ROWS=$2 DIM=$3 Q=$4 VARIANTS=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr_$1 -o tr -- python $R/tools/probes/r5_i8_variants.py > /tmp/tr_$1.log 2>&1
  F=$(find /tmp/tr_$1 -name "*kernel_stats.csv" | head -1)
  [ -n "$F" ] && cp $F $R/gpurun_out/r6_vector_leg_$1_kernel_stats_e.csv && head -12 $F | cut -c1-200
  grep -a "variant 0\|f32 level" /tmp/tr_$1.log | tee -a $R/gpurun_out/r6_vector_leg_lines_e.log
done
cd $R
c5() {
  label="$1"; shift
  env "$@" timeout 900 python bench.py --config c5 --no-pmc $C5_EXTRA 2>gpurun_out/r6_c5_$label.err | tail -1 > gpurun_out/r6_c5_$label.json
  python - "$label" <<'PY'
import json, sys
lab = sys.argv[1]
try:
    d = json.loads(open(f"gpurun_out/r6_c5_{lab}.json").read())
except Exception as e:
    print(lab, "no line:", e); sys.exit(0)
det = d.get("detail")
if det:
    try:
        d = json.load(open(det))
    except Exception:
        pass
for k, v in (d.get("densities") or {}).items():
    print(lab, k, {x: v.get(x) for x in ("value", "ms_per_step", "knn_only_ms_per_step", "bytes_streamed_over_allowed_row_bytes", "scan_share_of_the_step", "inexact_queries_last_step")},
          "scan ms", v["roofline"].get("avg_launch_ms"), "frac", v["roofline"].get("frac"), "items", v.get("items"), "parity", (v.get("parity") or {}).get("mismatches"))
if not d.get("densities"):
    print(lab, json.dumps(d)[:1500])
PY
}
{
  C5_EXTRA="" c5 rows_with_parity
  C5_EXTRA="--no-cpu-baseline"
  c5 rows
  c5 tiles MSI_VS_GATHER_PCT=1600
  c5 rows_no_i8 MSI_VS_I8_BF16=0
  c5 tiles_no_i8 MSI_VS_GATHER_PCT=1600 MSI_VS_I8_BF16=0
} 2>&1 | grep -v "^+" | tee gpurun_out/r6_c5_rows.log
# the C5 step's kernels with the gather
cd /tmp
rm -rf /tmp/tr_c5
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr_c5 -o tr -- python $R/bench.py --config c5 --no-pmc --no-cpu-baseline > /tmp/tr_c5.log 2>&1
F=$(find /tmp/tr_c5 -name "*kernel_stats.csv" | head -1)
[ -n "$F" ] && cp $F $R/gpurun_out/r6_bench_c5_kernel_stats.csv && head -16 $F | cut -c1-220
