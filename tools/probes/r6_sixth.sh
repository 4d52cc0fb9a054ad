# Round 6, sixth device call: the vector scan inside the keyword leg's tail (--legs tail) at three starting points
set -x
mkdir -p gpurun_out
for at in 0.6 0.75 0.9; do
  timeout 500 python bench.py --legs tail --tail-at $at --steps 20 --warmup 3 --no-cpu-baseline --no-also --no-pmc --kw-features 0 2>/dev/null | tail -1 > gpurun_out/r6_tail_$at.json
  python - <<PY
import json
d=json.loads(open('gpurun_out/r6_tail_$at.json').readline())
print('tail-at $at', 'value', d['value'], 'ms_per_step', d['ms_per_step'], 'p50', d.get('p50_latency_ms'), 'sweep ms', d['roofline']['avg_launch_ms'], 'kw only', d['legs'].get('keyword_only_queries_per_s'))
PY
done > gpurun_out/r6_tail.log 2>&1
cat gpurun_out/r6_tail.log
