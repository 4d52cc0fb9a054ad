# Round 6, third device call: (1) kernel traces of the vector leg alone at C4's and C2's shapes (what is left outside the sweeps);
# (2) the keyword leg by callers and by the combiner's batching wait now that it is not host-bound; (3) legs serial vs overlapped
set -x
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for shape in "c4 10000000 768 768" "c2 1000000 384 256"; do
  set -- $shape
  rm -rf /tmp/tr_$1
  N_ROWS=$2 DIM=$3 Q=$4 VARIANTS=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr_$1 -o tr -- python $R/tools/probes/r5_i8_variants.py > /tmp/tr_$1.log 2>&1
  F=$(find /tmp/tr_$1 -name "*kernel_stats.csv" | head -1)
  [ -n "$F" ] && cp $F $R/gpurun_out/r6_vector_leg_$1_kernel_stats.csv && head -14 $F | cut -c1-230
  grep -v amdgpu.ids /tmp/tr_$1.log | tail -3
done
cd $R
MSI_SEARCH_CPU_PROFILE=1 timeout 900 python tools/kw_leg.py --callers 512 --queries 3072 --segment 3072 --sweep "0:256,0:320,0:384,0:512" 2>&1 | grep -v amdgpu.ids | tail -4 > gpurun_out/r6_kw_callers.log
cat gpurun_out/r6_kw_callers.log | cut -c1-900
for w in 100 50; do
  MSI_VM_BATCH_WAIT_US=$w MSI_SEARCH_CPU_PROFILE=1 timeout 900 python tools/kw_leg.py --callers 384 --queries 3072 --segment 3072 --sweep "0:256,0:384" 2>&1 | grep -v amdgpu.ids | tail -2 | sed "s/^/batch_wait_us=$w /"
done > gpurun_out/r6_kw_batch_wait.log 2>&1
cat gpurun_out/r6_kw_batch_wait.log | cut -c1-700
for legs in serial overlap; do
  timeout 500 python bench.py --legs $legs --steps 20 --warmup 3 --no-cpu-baseline --no-also --no-pmc 2>/dev/null | tail -1 > gpurun_out/r6_legs2_$legs.json
  python - <<PY
import json
d=json.loads(open('gpurun_out/r6_legs2_$legs.json').readline())
print('$legs', 'value', d['value'], 'ms_per_step', d['ms_per_step'], 'p50', d.get('p50_latency_ms'), 'sweep ms', d['roofline']['avg_launch_ms'], 'legs', d.get('legs'))
PY
done > gpurun_out/r6_overlap2.log 2>&1
cat gpurun_out/r6_overlap2.log
