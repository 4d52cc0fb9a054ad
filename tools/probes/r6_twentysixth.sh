# Round 6, twenty-sixth device call: the keyword leg's counters on the final tree (bench.py --kw-roofline: children of the run
# under rocprofv3 --pmc, FETCH_SIZE / WRITE_SIZE / L2 hits over every vm_kernel dispatch) — HBM bytes per query against round
# 5's 141 MB (52.8 read + 87.9 written) and against the 18.6 MB of stored postings a fresh query reads
set -x
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
export TMPDIR=/tmp
ulimit -c 0
cd $R
( time timeout 1300 python bench.py --kw-roofline --no-also --no-cpu-baseline --steps 6 --warmup 2 --kw-features 0 --no-overlapped-leg 2>gpurun_out/r6_keyword_roofline.err | tail -1 > gpurun_out/r6_keyword_roofline_line.json ) 2>&1 | tail -3
python - <<'PY'
import json
d = json.load(open("gpurun_out/bench_detail_c4_n1.json"))
out = {"keyword_roofline": d.get("keyword_roofline"), "value": d.get("value"), "ms_per_step": d.get("ms_per_step"),
       "command": "python bench.py --kw-roofline --no-also --no-cpu-baseline --steps 6 --warmup 2 --kw-features 0 --no-overlapped-leg"}
json.dump(out, open("gpurun_out/r6_keyword_roofline.json", "w"))
print(json.dumps(out)[:3000])
PY
tail -3 gpurun_out/r6_keyword_roofline.err | cut -c1-300
