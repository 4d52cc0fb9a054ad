# Round 6, first device call: (1) where the keyword leg's host CPU goes on the fresh stream (final tree of round 5);
# (2) the two legs serial vs overlapped on one box
set -x
mkdir -p gpurun_out
nproc; rocm-smi --showmeminfo vram 2>/dev/null | head -5
RB_PROFILE_PER_THREAD=1 KW_PROFILE=gpurun_out/r6_kw_fresh.prof MSI_SEARCH_CPU_PROFILE=1 timeout 600 python tools/kw_leg.py --callers 256 --queries 3072 --fresh 4608 2>&1 | grep -v amdgpu.ids | tail -2 > gpurun_out/r6_kw_fresh_line.txt
python tools/r3_symbolize.py gpurun_out/r6_kw_fresh.prof 80 > gpurun_out/r6_kw_fresh_profile.txt 2>&1
rm -f gpurun_out/r6_kw_fresh.prof
for legs in serial overlap; do
  echo "== legs $legs"
  timeout 500 python bench.py --legs $legs --steps 20 --warmup 3 --no-cpu-baseline --no-also --no-pmc 2>/dev/null | tail -1 > gpurun_out/r6_legs_$legs.json
  python - <<PY
import json
d=json.loads(open('gpurun_out/r6_legs_$legs.json').readline())
print('$legs', 'value', d['value'], 'ms_per_step', d['ms_per_step'], 'p50', d.get('p50_latency_ms'), 'roofline', d['roofline'], 'legs', d.get('legs'))
PY
done > gpurun_out/r6_overlap.log 2>&1
cat gpurun_out/r6_overlap.log
