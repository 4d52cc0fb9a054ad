run() { echo "== $*"; env "$@" python bench.py --no-pmc --no-cpu-baseline --steps 8 --warmup 3 $EXTRA 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], 'frac', d['roofline']['frac'], 'kw_only', d['legs'].get('keyword_only_queries_per_s'), 'cpus', d['legs'].get('keyword_only_host_cpus_used'), 'lists/round', d['legs'].get('keyword_lists_per_launch_round'))"; }
EXTRA="" run A=1
EXTRA="" run MSI_SCAN_STREAM_PRIORITY=0
EXTRA="--kw-threads 64" run A=1
EXTRA="--kw-threads 192" run A=1
EXTRA="--serial-legs" run A=1
