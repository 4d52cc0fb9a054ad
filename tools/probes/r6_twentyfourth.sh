# Round 6, twenty-fourth device call: C4's rows leave HBM once the store holds them (the full-size parity check draws them again
# from the generator, synth.device_rows_chunks) — the default command as it is (256 callers x 512 slots) and with 384 callers x
# 448 slots, each with the free HBM at its fullest moments (config.hbm_free_gb_min) and its parity counts
set -x
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
export TMPDIR=/tmp
ulimit -c 0
cd $R
timeout 600 python -m pytest -q -m gpu tests/test_synth_rows_cpu.py 2>&1 | tail -2 | tee gpurun_out/r6_twentyfourth_tests.log
one() {
  tag=$1; shift
  ( time env "$@" timeout 900 python bench.py $EXTRA 2>gpurun_out/r6_bench_$tag.err | tail -1 > gpurun_out/r6_bench_$tag.json ) 2>&1 | tail -3
  cp gpurun_out/bench_detail_c4_n1.json gpurun_out/r6_bench_${tag}_detail.json
  python - "$tag" <<'PY'
import json, sys
tag = sys.argv[1]
try:
    d = json.loads(open(f"gpurun_out/r6_bench_{tag}.json").read())
    l = d.get("legs", {})
    print(tag, "| value", d["value"], "ms_per_step", d["ms_per_step"], "hbm free min GB", d["config"].get("hbm_free_gb_min"), "callers", d["config"].get("keyword_callers_per_rank"),
          "parity", json.dumps(d.get("parity")), "step parts", l.get("step_parts_ms"), "keyword_only", l.get("keyword_only_queries_per_s"), "lists", l.get("keyword_lists_per_query"),
          "side by side", json.dumps(l.get("legs_side_by_side")), "features", l.get("keyword_with_features_queries_per_s"), l.get("keyword_with_features_parity"), "seconds", d.get("seconds"), "bytes", len(json.dumps(d, separators=(",", ":"))))
except Exception as e:
    print(tag, "| FAILED", e)
    print(open(f"gpurun_out/r6_bench_{tag}.err").read()[-1500:])
PY
}
{
  EXTRA=""; one 256x512_rows_freed
  EXTRA="--kw-threads 384 --kw-slots 448"; one 384x448_rows_freed MSI_BENCH_CALLERS_PER_CPU=24
} 2>&1 | grep -v "^+" | tee gpurun_out/r6_rows_freed.log
