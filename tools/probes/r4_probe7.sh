# round 4, seventh GPU pass: callers of the keyword leg on the coherent corpus (its searches cost less host CPU than the
# round-3 workload's: is the leg latency-bound at 160 callers?)
set -x
mkdir -p gpurun_out
for T in 160 224 288; do
  MSI_BENCH_CALLERS_PER_CPU=20 timeout 600 python bench.py --gpus 1 --steps 10 --warmup 3 --no-also --no-pmc --no-cpu-baseline --kw-threads $T > gpurun_out/r4_callers_$T.json 2> gpurun_out/r4_callers_$T.err
  python - "$T" <<'PY'
import json,sys
d=json.load(open("gpurun_out/r4_callers_%s.json"%sys.argv[1]))
print(sys.argv[1], "value", d["value"], "ms/step", d["ms_per_step"], "callers", d["config"]["keyword_callers_per_rank"], "legs", {k:v for k,v in d["legs"].items() if k in ("vector_only_queries_per_s","keyword_only_queries_per_s","keyword_only_host_cpus_used","keyword_cold_posting_cache_queries_per_s")}, "lat", {k:v for k,v in d["latency"].items() if "ms" in k})
PY
done
