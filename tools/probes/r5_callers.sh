# Round 5: keyword caller threads now that a fresh query costs 0.86 ms of host CPU (9 of 16 CPUs at 160 callers)
for c in 208 256; do
  echo "== callers $c"
  MSI_BENCH_CALLERS_PER_CPU=16 timeout 400 python bench.py --kw-threads $c --steps 10 --warmup 3 --no-cpu-baseline --no-also --no-pmc 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('value', d['value'], 'ms_per_step', d['ms_per_step'], 'p50', d['p50_latency_ms'], 'callers', d['config']['keyword_callers_per_rank'], 'legs', d.get('legs'), d.get('latency_ms'))"
done
