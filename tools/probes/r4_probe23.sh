mkdir -p gpurun_out
timeout 400 python -m pytest tests/test_zz_vm_gpu.py tests/test_search_gpu.py tests/test_configs_gpu.py -m gpu -q -x 2>&1 | grep -E "passed|failed|error" | tail -3
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r4_bench_c4_final3.json 2> gpurun_out/r4_bench_c4_final3.err; echo bench rc=$?
python - <<'PY'
import json
d=json.load(open("gpurun_out/r4_bench_c4_final3.json"))
print("value", d["value"], "ms/step", d["ms_per_step"], "roofline", d["roofline"]["frac"], d["roofline"]["traffic"], "parity", d["parity"]["mismatches"])
print("legs", {k:v for k,v in d["legs"].items() if not isinstance(v, dict)})
a=d.get("also",{})
for k in ("c2","c3","c5"):
    v=a.get(k,{})
    print(k, {kk:v.get(kk) for kk in ("value","ms_per_step","error")}, "frac", (v.get("roofline") or {}).get("frac"), "traffic", (v.get("roofline") or {}).get("traffic"), "parity", (v.get("parity") or {}).get("mismatches"))
PY
