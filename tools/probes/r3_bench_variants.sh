run() { echo "== $*"; python bench.py --no-pmc --no-cpu-baseline --no-also --steps 10 --warmup 3 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], 'frac', d['roofline']['frac'], 'kw_only', d['legs'].get('keyword_only_queries_per_s'), 'cpus', d['legs'].get('keyword_only_host_cpus_used'))"; }
run --legs serial
run --legs tail --tail-at 0.9
run --legs tail --tail-at 0.8
run --legs tail --tail-at 0.7
run --legs tail --tail-at 0.6
