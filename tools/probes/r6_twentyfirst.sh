# Round 6, twenty-first device call — the final tree, as the driver will run it: the whole GPU tier, smoke(), the default
# command (its detail kept: legs.step_parts_ms, C5's per-step times), the rocprofv3 kernel trace of the C4 step of the same
# command; and C5's slow step looked at from the search side (MSI_SEARCH_DEBUG: searches done again for want of pool slots)
set -x
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
export TMPDIR=/tmp
ulimit -c 0
cd $R
( time timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 ) > gpurun_out/r6_gpu_tier_final.log 2>&1
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -a -v amdgpu.ids | tail -1 >> gpurun_out/r6_gpu_tier_final.log
cat gpurun_out/r6_gpu_tier_final.log
( time timeout 900 python bench.py 2>gpurun_out/r6_bench_default_final3.err | tail -1 > gpurun_out/r6_bench_default_final3.json ) 2>&1 | tail -3
cp gpurun_out/bench_detail_c4_n1.json gpurun_out/r6_bench_default_final3_detail.json
cut -c1-4200 gpurun_out/r6_bench_default_final3.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/r6_bench_default_final3_detail.json"))
print("step parts", json.dumps(d["legs"].get("step_parts_ms")))
for dens, line in ((d.get("also") or {}).get("c5") or {}).get("densities", {}).items():
    print("c5 in the default command | density", dens, "mean ms", line.get("ms_per_step"), "p50 ms", line.get("p50_latency_ms"), "steps", line.get("step_ms"))
PY
cd /tmp
rm -rf /tmp/tr_c4
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr_c4 -o tr -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-also --no-pmc --kw-features 0 --no-overlapped-leg > /tmp/tr_c4.log 2>&1
F=$(find /tmp/tr_c4 -name "*kernel_stats.csv" | head -1)
[ -n "$F" ] && cp $F $R/gpurun_out/r6_bench_c4_kernel_stats.csv && head -8 $F | cut -c1-220
tail -1 /tmp/tr_c4.log | cut -c1-4200 > $R/gpurun_out/r6_bench_c4_traced_line.json
cd $R
MSI_SEARCH_DEBUG=1 MSI_BENCH_DETAIL_DIR=/tmp/c5d timeout 600 python bench.py --config c5 --steps 6 --warmup 2 --no-cpu-baseline --no-pmc > /tmp/c5.log 2> /tmp/c5.err
{
  echo "searches done again for want of pool slots: $(grep -a -c 'ran out of pool slots' /tmp/c5.err)"
  grep -a "\[msi\]" /tmp/c5.err | sort | uniq -c | sort -rn | head -8
  python - <<'PY'
import json, glob
for f in glob.glob("/tmp/c5d/*.json"):
    d = json.load(open(f))
    for dens, line in (d.get("densities") or {}).items():
        print("c5, 6 steps behind 2 warm-up steps | density", dens, "mean ms", line.get("ms_per_step"), "p50 ms", line.get("p50_latency_ms"), "steps", line.get("step_ms"))
PY
} 2>&1 | tee gpurun_out/r6_c5_slow_step.log
