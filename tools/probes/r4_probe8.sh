set -x
mkdir -p gpurun_out
timeout 600 python -m pytest "tests/test_configs_gpu.py::test_rerank_inside_candidate_universes_on_the_corpus" -m gpu -q -x 2>&1 | tail -3
t0=$(date +%s)
timeout 900 python bench.py --config c5 --steps 5 --warmup 2 --no-pmc > gpurun_out/r4_c5_probe8.json 2> gpurun_out/r4_c5_probe8.err; echo rc=$?
echo "c5 took $(( $(date +%s) - t0 )) s"
python - <<'PY'
import json
d=json.load(open("gpurun_out/r4_c5_probe8.json"))
print("value", d["value"], "ms/step", d["ms_per_step"], "frac", d["roofline"]["frac"])
for k,v in d["densities"].items():
    print(k, {kk:v.get(kk) for kk in ("value","ms_per_step","knn_only_ms_per_step","words_typo_fast_path_queries_per_s","bytes_streamed_over_allowed_row_bytes","scan_share_of_the_step")}, "frac", v["roofline"]["frac"], "parity", v.get("parity",{}).get("mismatches"), "rerank", (v.get("parity",{}).get("rerank") or {}).get("mismatches"), (v.get("parity",{}).get("rerank") or {}).get("hits_compared"))
print(json.dumps(d.get("parity"))[:500])
PY
tail -3 gpurun_out/r4_c5_probe8.err
