# Round 6, twenty-second device call: (1) the search tests on the library with the checked rank-table allocation;
# (2) why the keyword leg takes 53.5 ms inside the serial step and 47.2 ms on its own (legs.keyword_gap_probe: the leg behind
# 12 ms of idle time, behind the vector leg); (3) LAST, because it may fault: the run that dumped a GPU core in the nineteenth
# call (384 callers x 448 slots beside the C4 store exhaust HBM in the middle of the searches) must now end with MSI_E_OOM
set -x
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
export TMPDIR=/tmp
ulimit -c 0
cd $R
( time timeout 1200 python -m pytest -x -q -m gpu tests/test_search_gpu.py tests/test_zz_vm_gpu.py "tests/test_configs_gpu.py::test_c4_keyword_leg_on_the_coherent_corpus" 2>&1 | tail -4 ) > gpurun_out/r6_twentysecond_tests.log 2>&1
cat gpurun_out/r6_twentysecond_tests.log
for i in 1 2; do
timeout 900 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-also --no-pmc --kw-features 0 --kw-gap-probe 2>/dev/null | grep -a "^{" | tail -1 | python -c '
import sys, json
d = json.loads(sys.stdin.readline())
l = d.get("legs", {})
print("value", d["value"], "ms_per_step", d["ms_per_step"], "step parts", l.get("step_parts_ms"), "keyword_only", l.get("keyword_only_queries_per_s"), "side by side", json.dumps(l.get("legs_side_by_side")))'
python - <<'PY'
import json
d = json.load(open("gpurun_out/bench_detail_c4_n1.json"))
print("gap probe", json.dumps(d["legs"].get("keyword_gap_probe")))
PY
done 2>&1 | grep -v "^+" | tee gpurun_out/r6_keyword_gap_probe.log
MSI_BENCH_CALLERS_PER_CPU=64 timeout 900 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-also --no-pmc --kw-features 0 --no-overlapped-leg --kw-threads 384 --kw-slots 448 > /tmp/oom.log 2>&1
echo "exit code of the run that exhausts HBM: $?" | tee gpurun_out/r6_hbm_exhausted.log
grep -a "MSI_E_\|failed with\|Error\|core" /tmp/oom.log | sort | uniq -c | sort -rn | head -6 | cut -c1-300 | tee -a gpurun_out/r6_hbm_exhausted.log
