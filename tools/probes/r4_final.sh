# round-4 validation pass on the MI355X box: the GPU tier, smoke, the driver's own bench command, a kernel-trace summary of the
# bench INCLUDING the `also` legs (C2 / C3 / C5 / clustered)
set -x
mkdir -p gpurun_out
t0=$(date +%s)
timeout 1100 python -m pytest tests -m gpu -q -x 2>&1 | tail -6
echo "tests took $(( $(date +%s) - t0 )) s"
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
t0=$(date +%s)
timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r4_bench_c4_final2.json 2> gpurun_out/r4_bench_c4_final2.err; echo bench rc=$?
echo "bench took $(( $(date +%s) - t0 )) s"
python - <<'PY'
import json
d=json.load(open("gpurun_out/r4_bench_c4_final2.json"))
print("value", d["value"], "ms/step", d["ms_per_step"], "roofline", d["roofline"]["frac"], d["roofline"]["traffic"], "parity", d["parity"]["mismatches"])
print("legs", {k:v for k,v in d["legs"].items() if not isinstance(v, dict)})
a=d.get("also",{})
for k in ("c2","c3","c5"):
    v=a.get(k,{})
    print(k, {kk:v.get(kk) for kk in ("value","ms_per_step","seconds","error")}, "frac", (v.get("roofline") or {}).get("frac"), "parity", (v.get("parity") or {}).get("mismatches"))
print("clustered", {k:(v.get("value"), (v.get("roofline") or {}).get("frac"), (v.get("parity") or {}).get("mismatches"), v.get("error")) for k,v in a.get("clustered",{}).items()})
PY
tail -3 gpurun_out/r4_bench_c4_final2.err
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r4_prof_bench2 -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-pmc > $GRAFT_REPO_ROOT/gpurun_out/r4_prof_bench2.log 2>&1; echo rocprof rc=$?
cd $GRAFT_REPO_ROOT; f=$(find gpurun_out/r4_prof_bench2 -name "*kernel_stats.csv" | head -1); echo $f; cp "$f" gpurun_out/r4_bench_kernel_stats2.csv; head -16 "$f" | cut -c1-200
