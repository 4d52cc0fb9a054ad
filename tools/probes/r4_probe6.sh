# round 4, sixth GPU pass: the coherent-corpus keyword leg (parity test at 10 M documents, then the default bench line on it),
# the vector store's levels of effort on the device (updated tests, clustered legs inside the bench)
set -x
mkdir -p gpurun_out
t0=$(date +%s)
timeout 900 python -m pytest tests/test_vs_gpu.py "tests/test_configs_gpu.py::test_c4_keyword_leg_on_the_coherent_corpus" -m gpu -q -x 2>&1 | tail -5
echo "tests took $(( $(date +%s) - t0 )) s"
t0=$(date +%s)
timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r4_bench_c4_probe6.json 2> gpurun_out/r4_bench_c4_probe6.err; echo bench rc=$?
echo "bench took $(( $(date +%s) - t0 )) s"
python - <<'PY'
import json
d=json.load(open("gpurun_out/r4_bench_c4_probe6.json"))
print("value", d["value"], "ms/step", d["ms_per_step"], "roofline", d["roofline"]["frac"], "setup", d["config"]["setup_seconds"])
print("legs", json.dumps(d["legs"])[:2500])
print("latency", json.dumps(d["latency"])[:600])
print("cpu_baseline", json.dumps({k:v for k,v in d.get("cpu_baseline",{}).items() if k in ("value","keyword_queries_per_s")}))
p=d.get("parity",{}); print("parity", p.get("mismatches"), json.dumps(p.get("keyword"))[:600])
kr=d.get("keyword_roofline",{}); print("keyword_roofline", json.dumps({k:kr.get(k) for k in ("rounds","host_cpu_per_query","l2_counters","hbm_traffic_mb_per_query","hbm_frac","l2_request_frac","child_queries_per_s")})[:1800])
a=d.get("also",{})
for k in ("c2","c3","c5"):
    v=a.get(k,{})
    print(k, json.dumps({kk:v.get(kk) for kk in ("value","ms_per_step","seconds","error")}), "frac", (v.get("roofline") or {}).get("frac"), "parity", (v.get("parity") or {}).get("mismatches"))
print("clustered", json.dumps({k:{kk:v.get(kk) for kk in ("value","resolved","seconds","error")} | {"frac": (v.get("roofline") or {}).get("frac"), "e2e": (v.get("roofline") or {}).get("end_to_end_frac"), "parity": (v.get("parity") or {}).get("mismatches")} for k,v in a.get("clustered",{}).items()}))
PY
tail -5 gpurun_out/r4_bench_c4_probe6.err
