run() { echo "== $*"; env "$@" python bench.py --no-pmc --no-cpu-baseline --steps 8 --warmup 3 $EXTRA 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], 'frac', d['roofline']['frac'], 'kw_only', d['legs'].get('keyword_only_queries_per_s'), 'cpus', d['legs'].get('keyword_only_host_cpus_used'), 'lists/round', d['legs'].get('keyword_lists_per_launch_round'))"; }
EXTRA="" run GPU_MAX_HW_QUEUES=4
EXTRA="" run GPU_MAX_HW_QUEUES=8
EXTRA="--kw-threads 128" run GPU_MAX_HW_QUEUES=8
EXTRA="--kw-threads 192" run GPU_MAX_HW_QUEUES=8
EXTRA="" run GPU_MAX_HW_QUEUES=8 MSI_SEARCH_COMPACT=0
