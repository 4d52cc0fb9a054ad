# round 4, fifth GPU pass: can the two legs of the hybrid step share the device?  The 96-query scan (NQT = 6) holds 226 VGPRs x 2
# waves per SIMD and 147 KB of LDS per CU: no command-list workgroup fits beside it.  With 64 queries per sweep (NQT = 4: 166
# VGPRs, 98 KB) one does.  Serial vs overlapped legs at 6 / 5 / 4 query tiles per sweep.
set -x
mkdir -p gpurun_out
run() {  # name, env, legs
  env $2 timeout 600 python bench.py --gpus 1 --steps 10 --warmup 3 --no-also --no-pmc --no-cpu-baseline --legs $3 > gpurun_out/r4_overlap_$1.json 2> gpurun_out/r4_overlap_$1.err
  python - "$1" <<'PY'
import json,sys
d=json.load(open("gpurun_out/r4_overlap_%s.json"%sys.argv[1]))
print(sys.argv[1], "value", d["value"], "ms/step", d["ms_per_step"], "scan frac", d["roofline"]["frac"], "scan ms", d["roofline"]["avg_launch_ms"], "q/sweep", d["config"]["queries_per_hbm_sweep"], "legs", {k:v for k,v in d["legs"].items() if k in ("vector_only_queries_per_s","keyword_only_queries_per_s","keyword_only_host_cpus_used")})
PY
}
run serial6 MSI_X=1 serial
run overlap6 MSI_X=1 overlap
run overlap4 MSI_VS_MAX_QUERY_TILES=4 overlap
run overlap5 MSI_VS_MAX_QUERY_TILES=5 overlap
run serial4 MSI_VS_MAX_QUERY_TILES=4 serial
