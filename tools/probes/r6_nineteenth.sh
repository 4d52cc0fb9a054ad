# Round 6, nineteenth device call: MORE SEARCHES IN FLIGHT with SMALLER POOLS.  A step is 768 searches; on 256 callers that is
# three generations and a tail (keyword leg in the step 13.9 k q/s against 17-18 k in a continuous stream).  512 callers did
# not fit HBM with pools of 512 slots (640 MB each at 10 M documents) — but most searches continue in a compact space and use
# a fraction of their pool.  The emulator says what fewer slots cost: lists per query 11.1 / 12.5 / 17.6 / 24.0 at 512 / 384 /
# 256 / 192 slots (the bucket sort's tasks are admitted by free slots) — so only mild trades are worth a device run:
# callers x slots 256 x 512 (today), 320 x 512, 384 x 448 — the leg on its own in a continuous stream (tools/kw_leg.py),
# then the hybrid step (bench.py, C4 only).
set -x
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
export TMPDIR=/tmp
cd $R
run() {
  label="$1"; callers=$2; slots=$3; shift 3
  env "$@" timeout 900 python tools/kw_leg.py --callers $callers --slots $slots --queries 3072 --fresh 4608 2>&1 | grep -a -v amdgpu.ids | tail -1 | python -c '
import sys, json
line = sys.stdin.readline()
try:
    d = json.loads(line)
    print(sys.argv[1], "|", d["queries_per_s"], "q/s, host CPUs", d["host_cpus_used"], "p50", d["p50_ms_at_load"], "vm", json.dumps(d["vm"]), "cpu us/query", d.get("host_cpu_us_per_query", {}).get("search_threads"))
except Exception as e:
    print(sys.argv[1], "| FAILED:", line[:300])' "$label"
}
{
  run "256 callers x 512 slots, companion pool of 1024 slots (new default)" 256 512
  run "256 callers x 512 slots, companion pool of 512 slots (before)" 256 512 MSI_BITS_COMPANION_SLOTS_X=1
  run "384 callers x 448 slots" 384 448
  run "256 callers x 512 slots, companion pool of 1024 slots, again" 256 512
} 2>&1 | grep -v "^+\|^import\|^d = \|^print\|^line\|^try\|^except\|^    " | tee gpurun_out/r6_callers_and_slots.log
c4() {
  label="$1"; callers=$2; slots=$3; shift 3
  env MSI_BENCH_CALLERS_PER_CPU=64 "$@" timeout 900 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-also --no-pmc --kw-features 0 --kw-threads $callers --kw-slots $slots 2>/tmp/c4.err | tail -1 | python -c '
import sys, json
line = sys.stdin.readline()
try:
    d = json.loads(line)
    l = d.get("legs", {})
    print(sys.argv[1], "| value", d["value"], "ms_per_step", d["ms_per_step"], "p50", d.get("p50_latency_ms"), "scan ms", d["roofline"].get("avg_launch_ms"), "frac", d["roofline"].get("frac"), "vector_only", l.get("vector_only_queries_per_s"), "keyword_only", l.get("keyword_only_queries_per_s"), "cpus", l.get("keyword_only_host_cpus_used"), "lists", l.get("keyword_lists_per_query"), "side by side", json.dumps(l.get("legs_side_by_side")), "parity", json.dumps(d.get("parity")))
except Exception as e:
    print(sys.argv[1], "| FAILED:", line[:300])' "$label"
  tail -2 /tmp/c4.err | cut -c1-300
}
{
  c4 "step, 384 callers x 448 slots" 384 448
  c4 "step, 320 callers x 512 slots" 320 512
  c4 "step, 256 callers x 512 slots" 256 512
} 2>&1 | grep -v "^+\|^import\|^d = \|^l = \|^print\|^line\|^try\|^except\|^    " | tee gpurun_out/r6_step_callers_and_slots.log
# C5 inside the default command showed 99 ms per step at 10 % (standalone, 20 steps: 34-41 ms): 5 steps behind 2 warm-up steps —
# mean against median of the same five steps
MSI_BENCH_DETAIL_DIR=/tmp/c5d timeout 600 python bench.py --config c5 --steps 5 --warmup 2 --no-cpu-baseline --no-pmc > /tmp/c5.log 2>&1
python - <<'PY' | tee gpurun_out/r6_c5_five_steps.log
import json, glob
for f in glob.glob("/tmp/c5d/*c5*.json"):
    d = json.load(open(f))
    for dens, line in (d.get("densities") or {}).items():
        print("c5, 5 steps behind 2 warm-up steps | density", dens, "mean ms", line.get("ms_per_step"), "p50 ms", line.get("p50_latency_ms"), "knn only ms", line.get("knn_only_ms_per_step"), "q/s", line.get("value"))
PY
