# Round 6, fourteenth device call: what the thirteenth did not get to (bench.py had a syntax error in its C5 leg): C5 with the
# row-granular sweep / the tile-granular one / other gather thresholds, the bf16 store's int8 copy at k = 20, the C5 step's
# kernel trace, and the typo lookup's kernels at the batch the C4 step issues (1 536 words) and at C3's (8 192)
set -x
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest -x -q -m gpu tests/test_vs_gpu.py -k "row_granular or bf16" "tests/test_configs_gpu.py::test_c5_shard_bf16_filtered_k1000" 2>&1 | grep -a "passed\|failed\|error\|Error\|assert" | tail -8 | tee $R/gpurun_out/r6_fourteenth_tests.log
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
  c5 rows_gather_always MSI_VS_GATHER_PCT=0
  c5 rows_gather_400 MSI_VS_GATHER_PCT=400
} 2>&1 | grep -v "^+" | tee gpurun_out/r6_c5_rows.log
# an unfiltered bf16 store of C5's shard shape at k = 20: level 0 is the sweep of its int8 copy (half the bytes of the bf16 rows)
cd /tmp
N_ROWS=12500000 DIM=1024 Q=256 VARIANTS=0 STORAGE=bf16 timeout 600 python $R/tools/probes/r5_i8_variants.py 2>&1 | grep -a "variant 0\|f32 level" | sed "s/^/bf16 store, int8 copy: /" | tee -a $R/gpurun_out/r6_bf16_store_i8.log
# the C5 step's kernels with the gather
cd /tmp
rm -rf /tmp/tr_c5
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr_c5 -o tr -- python $R/bench.py --config c5 --no-pmc --no-cpu-baseline > /tmp/tr_c5.log 2>&1
F=$(find /tmp/tr_c5 -name "*kernel_stats.csv" | head -1)
[ -n "$F" ] && cp $F $R/gpurun_out/r6_bench_c5_kernel_stats.csv && head -16 $F | cut -c1-220
cd /tmp
for B in 1536 8192; do
  rm -rf /tmp/tr_c3_$B
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr_c3_$B -o tr -- python $R/bench.py --config c3 --queries $B --no-pmc --no-cpu-baseline > /tmp/tr_c3_$B.log 2>&1
  F=$(find /tmp/tr_c3_$B -name "*kernel_stats.csv" | head -1)
  [ -n "$F" ] && cp $F $R/gpurun_out/r6_c3_${B}_kernel_stats.csv && head -8 $F | cut -c1-200
  tail -1 /tmp/tr_c3_$B.log | cut -c1-400
done
