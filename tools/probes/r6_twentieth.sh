# Round 6, twentieth device call: the one-word searches' KNOWN OUTCOMES (Words / Proximity / single-level Typo evaluations
# answered without a round) and 12 levels per wait, on the device: the search tests, the keyword leg with and without, the
# default command on the final tree, C5 in five steps (the default command's extra showed 99 ms per step at 10 %).
# (the nineteenth call's 384 x 448 step exhausted HBM in the middle of its searches and dumped a GPU core that filled the box's
# disk: the two steps behind it failed on that, not on their own sizes)
set -x
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
export TMPDIR=/tmp
ulimit -c 0
export HSA_ENABLE_COREDUMP=0 HIP_ENABLE_COREDUMP=0
cd $R
( time timeout 1200 python -m pytest -x -q -m gpu tests/test_search_gpu.py tests/test_zz_vm_gpu.py tests/test_zz_levels_per_wait_gpu.py "tests/test_configs_gpu.py::test_c4_keyword_leg_on_the_coherent_corpus" "tests/test_configs_gpu.py::test_postings_staged_at_index_open_on_the_coherent_corpus" "tests/test_configs_gpu.py::test_phrases_on_the_coherent_corpus" 2>&1 | tail -5 ) > gpurun_out/r6_twentieth_tests.log 2>&1
cat gpurun_out/r6_twentieth_tests.log
run() {
  label="$1"; shift
  env "$@" timeout 900 python tools/kw_leg.py --callers 256 --queries 3072 --fresh 4608 2>&1 | grep -a -v amdgpu.ids | tail -1 | python -c '
import sys, json
line = sys.stdin.readline()
try:
    d = json.loads(line)
    print(sys.argv[1], "|", d["queries_per_s"], "q/s, host CPUs", d["host_cpus_used"], "p50", d["p50_ms_at_load"], "vm", json.dumps(d["vm"]))
except Exception as e:
    print(sys.argv[1], "| FAILED:", line[:300])' "$label"
}
{
  run "known outcomes, 12 levels per wait (new defaults)"
  run "no known outcomes, 8 levels per wait (before)" MSI_SEARCH_KNOWN_OUTCOMES=0 MSI_SEARCH_LEVELS_PER_WAIT=8
  run "known outcomes, 12 levels per wait, again"
} 2>&1 | grep -v "^+\|^import\|^d = \|^print\|^line\|^try\|^except\|^    " | tee gpurun_out/r6_known_outcomes.log
( time timeout 900 python bench.py 2>gpurun_out/r6_bench_default_final2.err | tail -1 > gpurun_out/r6_bench_default_final2.json ) 2>&1 | tail -3
cp gpurun_out/bench_detail_c4_n1.json gpurun_out/r6_bench_default_final2_detail.json
cut -c1-4200 gpurun_out/r6_bench_default_final2.json
MSI_BENCH_DETAIL_DIR=/tmp/c5d timeout 600 python bench.py --config c5 --steps 5 --warmup 2 --no-cpu-baseline --no-pmc > /tmp/c5.log 2>&1
tail -2 /tmp/c5.log | cut -c1-600
python - <<'PY' | tee gpurun_out/r6_c5_five_steps.log
import json, glob
for f in glob.glob("/tmp/c5d/*.json"):
    d = json.load(open(f))
    for dens, line in (d.get("densities") or {}).items():
        print("c5, 5 steps behind 2 warm-up steps | density", dens, "mean ms", line.get("ms_per_step"), "p50 ms", line.get("p50_latency_ms"), "knn only ms", line.get("knn_only_ms_per_step"), "q/s", line.get("value"))
PY
