# Round 6, seventeenth device call: FEWER hardware queues / streams for the keyword rounds (the twelfth call measured 24 and 32
# queues at 0.57 / 0.37 of 16: is 16 itself past the optimum?), and two knobs of the search that were never measured on this
# tree (MSI_SEARCH_LATE_WAIT, MSI_SEARCH_LEVELS_PER_WAIT) — tools/kw_leg.py, 256 callers, fresh queries, each its own process
set -x
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
export TMPDIR=/tmp
cd $R
run() {
  label="$1"; shift
  env "$@" timeout 900 python tools/kw_leg.py --callers 256 --queries 3072 --fresh 4608 2>&1 | grep -a -v amdgpu.ids | tail -1 | python -c '
import sys, json
d = json.loads(sys.stdin.readline())
print(sys.argv[1], "|", d["queries_per_s"], "q/s, host CPUs", d["host_cpus_used"], "p50", d["p50_ms_at_load"], "vm", json.dumps(d["vm"]))' "$label"
}
{
  run "16 queues, 16 streams"
  run "12 queues, 12 streams" GPU_MAX_HW_QUEUES=12 MSI_VM_STREAMS=12
  run "8 queues, 8 streams" GPU_MAX_HW_QUEUES=8 MSI_VM_STREAMS=8
  run "8 queues, 16 streams" GPU_MAX_HW_QUEUES=8
  run "16 queues, 8 streams" MSI_VM_STREAMS=8
  run "20 queues, 16 streams" GPU_MAX_HW_QUEUES=20
  run "16 queues, 16 streams, late wait" MSI_SEARCH_LATE_WAIT=1
  run "16 queues, 16 streams, 12 levels per wait" MSI_SEARCH_LEVELS_PER_WAIT=12
  run "16 queues, 16 streams, again"
} 2>&1 | grep -v "^+\|^import\|^d = \|^print" | tee gpurun_out/r6_queues.log
# ... and the hybrid step's legs side by side with the vector sweep CONFINED to part of the CUs (MSI_SCAN_CUS: the scan's stream
# carries a CU mask, the sweep's grid follows): the keyword rounds always find free CUs, the sweeps take longer but hide
# inside the keyword leg
c4() {
  label="$1"; shift
  env "$@" timeout 900 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-also --no-pmc --kw-features 0 $C4_EXTRA 2>/dev/null | tail -1 | python -c '
import sys, json
d = json.loads(sys.stdin.readline())
l = d.get("legs", {})
print(sys.argv[1], "| value", d["value"], "ms_per_step", d["ms_per_step"], "p50", d.get("p50_latency_ms"), "scan ms", d["roofline"].get("avg_launch_ms"), "frac", d["roofline"].get("frac"), "vector_only", l.get("vector_only_queries_per_s"), "keyword_only", l.get("keyword_only_queries_per_s"), "cpus", l.get("keyword_only_host_cpus_used"))' "$label"
}
{
  C4_EXTRA="--legs overlap"
  c4 "overlap, sweep on 192 CUs" MSI_SCAN_CUS=192
  c4 "overlap, sweep on 128 CUs" MSI_SCAN_CUS=128
  c4 "overlap, sweep on 96 CUs" MSI_SCAN_CUS=96
  c4 "overlap, sweep on 128 CUs, x8 workgroups" MSI_SCAN_CUS=128 MSI_VS_GRID_MULT=8
  c4 "overlap, sweep on 192 CUs, x8 workgroups" MSI_SCAN_CUS=192 MSI_VS_GRID_MULT=8
  c4 "overlap x16" MSI_VS_GRID_MULT=16
  C4_EXTRA="--legs serial"; c4 "serial"
} 2>&1 | grep -v "^+\|^import\|^d = \|^l = \|^print" | tee gpurun_out/r6_overlap_cu_mask.log
