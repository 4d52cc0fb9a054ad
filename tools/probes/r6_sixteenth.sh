# Round 6, sixteenth device call: the hybrid step's legs side by side with the sweep cut into short workgroups (MSI_VS_GRID_MULT;
# no stream priorities: the fifteenth call measured them at 0.6 of the throughput in every order) — overlap and tail orders,
# several multipliers, each twice, one box; the typo lookup on a batch in dictionary order and its in-kernel phase timers
set -x
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
export TMPDIR=/tmp
cd $R
c4() {
  label="$1"; shift
  env "$@" timeout 900 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-also --no-pmc --kw-features 0 $C4_EXTRA 2>/dev/null | tail -1 | python -c '
import sys, json
d = json.loads(sys.stdin.readline())
l = d.get("legs", {})
print(sys.argv[1], "| value", d["value"], "ms_per_step", d["ms_per_step"], "p50", d.get("p50_latency_ms"), "scan ms", d["roofline"].get("avg_launch_ms"), "frac", d["roofline"].get("frac"), "vector_only", l.get("vector_only_queries_per_s"), "keyword_only", l.get("keyword_only_queries_per_s"), "cpus", l.get("keyword_only_host_cpus_used"))' "$label"
}
{
  for rep in 1 2; do
    C4_EXTRA="--legs serial"; c4 "serial ($rep)"
    C4_EXTRA="--legs overlap"; c4 "overlap x8 ($rep)" MSI_VS_GRID_MULT=8
    c4 "overlap x16 ($rep)" MSI_VS_GRID_MULT=16
    c4 "overlap x32 ($rep)" MSI_VS_GRID_MULT=32
    C4_EXTRA="--legs tail --tail-at 0.75"; c4 "tail at 0.75 x8 ($rep)" MSI_VS_GRID_MULT=8
    C4_EXTRA="--legs tail --tail-at 0.85"; c4 "tail at 0.85 x8 ($rep)" MSI_VS_GRID_MULT=8
  done
  C4_EXTRA="--legs tail --tail-at 0.75"; c4 "tail at 0.75 x1"
  C4_EXTRA="--legs overlap"; c4 "overlap x1"
} 2>&1 | grep -v "^+\|^import\|^d = \|^l = \|^print" | tee gpurun_out/r6_overlap_short_wgs.log
c3() {
  label="$1"; shift
  env "$@" timeout 600 python bench.py --config c3 --no-pmc --no-cpu-baseline $C3_EXTRA 2>gpurun_out/r6_c3_tmp.err | tail -1 | python -c '
import sys, json
d = json.loads(sys.stdin.readline())
print(sys.argv[1], "|", d["value"], d["unit"], "ms_per_step", d["ms_per_step"])' "$label"
  grep -a "msi_dict profile" gpurun_out/r6_c3_tmp.err | tail -1
}
{
  C3_EXTRA="--queries 8192"; c3 "8192"
  c3 "8192 batch in dictionary order" MSI_BENCH_C3_SORTED=1
  c3 "8192 in-kernel timers" MSI_DICT_PROFILE=1
  C3_EXTRA="--queries 1536"; c3 "1536"
  c3 "1536 batch in dictionary order" MSI_BENCH_C3_SORTED=1
  c3 "1536 in-kernel timers" MSI_DICT_PROFILE=1
  C3_EXTRA="--queries 32768"; c3 "32768"
} 2>&1 | grep -v "^+\|^import\|^d = \|^print" | tee gpurun_out/r6_c3_sorted.log
