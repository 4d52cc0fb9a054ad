# round 4, fourth GPU pass: the default bench line with the round's new objects; C3 at three batch sizes
set -x
mkdir -p gpurun_out
t0=$(date +%s)
timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r4_bench_c4_probe4.json 2> gpurun_out/r4_bench_c4_probe4.err; echo bench rc=$?
echo "bench took $(( $(date +%s) - t0 )) s"
python - <<'PY'
import json
d=json.load(open("gpurun_out/r4_bench_c4_probe4.json"))
print("value", d["value"], "ms/step", d["ms_per_step"], "roofline", d["roofline"]["frac"], "traffic", d["roofline"]["traffic"])
print("legs", json.dumps(d["legs"])[:900])
print("latency", json.dumps(d["latency"])[:600])
print("cpu_baseline", json.dumps(d.get("cpu_baseline"))[:900])
print("parity", json.dumps(d.get("parity"))[:1200])
print("keyword_roofline", json.dumps(d.get("keyword_roofline"))[:3000])
a=d.get("also",{})
for k in ("c2","c3","c5"):
    v=a.get(k,{})
    print(k, json.dumps({kk:v.get(kk) for kk in ("value","ms_per_step","seconds","error")}), json.dumps(v.get("roofline"))[:500], json.dumps(v.get("parity"))[:300])
print("c3 dict_roofline", json.dumps(a.get("c3",{}).get("dict_roofline"))[:1500])
print("c5 densities", json.dumps({k:{kk:v.get(kk) for kk in ("value","ms_per_step","knn_only_ms_per_step","bytes_streamed_over_allowed_row_bytes","scan_share_of_the_step")} | {"frac": v["roofline"]["frac"], "parity": v.get("parity",{}).get("mismatches")} for k,v in a.get("c5",{}).get("densities",{}).items()}))
print("clustered", json.dumps(a.get("clustered"))[:3000])
PY
tail -5 gpurun_out/r4_bench_c4_probe4.err
echo "== C3 batch sizes"
for Q in 1536 8192 32768; do timeout 300 python bench.py --config c3 --steps 10 --warmup 2 --no-pmc --no-cpu-baseline --queries $Q 2>/dev/null | cut -c1-260; done
