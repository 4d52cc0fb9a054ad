# Round 6, twenty-seventh device call: the rocprofv3 kernel trace of the C4 step of the default command on the final tree, with
# the line that run printed beside it (the roofline object's avg_launch_ms against the trace's average for the same kernel)
set -x
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
export TMPDIR=/tmp
ulimit -c 0
cd /tmp
rm -rf /tmp/tr_c4
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr_c4 -o tr -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-also --no-pmc --kw-features 0 --no-overlapped-leg > /tmp/tr_c4.log 2>&1
F=$(find /tmp/tr_c4 -name "*kernel_stats.csv" | head -1)
[ -n "$F" ] && cp $F $R/gpurun_out/r6_bench_c4_kernel_stats.csv && head -8 $F | cut -c1-220
grep -a "^{" /tmp/tr_c4.log | tail -1 > $R/gpurun_out/r6_bench_c4_traced_line.json
python - <<'PY'
import json, os
d = json.loads(open(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r6_bench_c4_traced_line.json").read())
print("traced run: value", d["value"], "ms_per_step", d["ms_per_step"], "roofline", json.dumps(d["roofline"]))
PY
