# Round 6, fifteenth device call: (1) the typo lookup with dict_other_kernel beside the range scans (DictArgs::defer_caps) at
# C3's batch and at the C4 step's, against the serial order, and the slice width at the small batch; (2) the hybrid step's two
# legs side by side with the keyword rounds' streams at the highest dispatch priority and the vector sweep cut into short
# workgroups (MSI_VM_STREAM_PRIORITY, MSI_VS_GRID_MULT) — each configuration its own process on one box
set -x
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest -x -q -m gpu tests/test_dict_gpu.py tests/test_zz_fst_gpu.py "tests/test_configs_gpu.py::test_c3_2m_term_dictionary" 2>&1 | grep -a "passed\|failed\|error\|Error\|assert" | tail -8 | tee gpurun_out/r6_fifteenth_tests.log
c3() {
  label="$1"; shift
  env "$@" timeout 600 python bench.py --config c3 --no-pmc --no-cpu-baseline $C3_EXTRA 2>/dev/null | tail -1 | python -c '
import sys, json
d = json.loads(sys.stdin.readline())
print(sys.argv[1], d["value"], d["unit"], "ms_per_step", d["ms_per_step"], "kernel ms", d.get("cache_stream", {}).get("avg_launch_ms") if isinstance(d.get("cache_stream"), dict) else None)' "$label"
}
{
  C3_EXTRA="--queries 8192"; c3 "8192 deferred"
  c3 "8192 serial" MSI_DICT_DEFER_CAPS_MIN=0
  C3_EXTRA="--queries 1536"; c3 "1536 deferred"
  c3 "1536 serial" MSI_DICT_DEFER_CAPS_MIN=0
  c3 "1536 deferred slices of 128 tiles" MSI_DICT_SLICE_TILES=128
  c3 "1536 deferred slices of 64 tiles" MSI_DICT_SLICE_TILES=64
  c3 "1536 deferred slices of 32 tiles" MSI_DICT_SLICE_TILES=32
  C3_EXTRA="--queries 8192"; c3 "8192 deferred slices of 128 tiles" MSI_DICT_SLICE_TILES=128
} 2>&1 | grep -v "^+" | tee gpurun_out/r6_c3_deferred.log
c4() {
  label="$1"; shift
  env "$@" timeout 900 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-also --no-pmc --kw-features 0 $C4_EXTRA 2>/dev/null | tail -1 | python -c '
import sys, json
d = json.loads(sys.stdin.readline())
l = d.get("legs", {})
print(sys.argv[1], "value", d["value"], "ms_per_step", d["ms_per_step"], "p50", d.get("p50_latency_ms"), "scan ms", d["roofline"].get("avg_launch_ms"), "vector_only", l.get("vector_only_queries_per_s"), "keyword_only", l.get("keyword_only_queries_per_s"), "cpus", l.get("keyword_only_host_cpus_used"), "alg", l.get("keyword_algorithmic_bytes_per_query"))' "$label"
}
{
  C4_EXTRA="--legs serial"; c4 "serial"
  C4_EXTRA="--legs overlap"; c4 "overlap"
  c4 "overlap, rounds at high priority, sweep x8 workgroups" MSI_VM_STREAM_PRIORITY=1 MSI_VS_GRID_MULT=8
  c4 "overlap, rounds at high priority, sweep x16 workgroups" MSI_VM_STREAM_PRIORITY=1 MSI_VS_GRID_MULT=16
  c4 "overlap, rounds at high priority, sweep x4 workgroups" MSI_VM_STREAM_PRIORITY=1 MSI_VS_GRID_MULT=4
  c4 "overlap, sweep x8 workgroups" MSI_VS_GRID_MULT=8
  c4 "overlap, rounds at high priority" MSI_VM_STREAM_PRIORITY=1
  C4_EXTRA="--legs serial"; c4 "serial, rounds at high priority" MSI_VM_STREAM_PRIORITY=1
  C4_EXTRA="--legs tail --tail-at 0.75"; c4 "tail at 0.75, rounds at high priority, sweep x8 workgroups" MSI_VM_STREAM_PRIORITY=1 MSI_VS_GRID_MULT=8
} 2>&1 | grep -v "^+" | tee gpurun_out/r6_overlap_priority.log
