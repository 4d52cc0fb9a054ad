# round 3: more keyword callers per step now that a query costs less host CPU
run() { echo "== $*"; python bench.py --no-pmc --no-cpu-baseline --no-also --steps 8 --warmup 3 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['value'], 'q/s', d['ms_per_step'], 'ms/step; scan frac', d['roofline']['frac'], '; kw_only', d['legs'].get('keyword_only_queries_per_s'), 'cpus', d['legs'].get('keyword_only_host_cpus_used'), 'lists/round', d['legs'].get('keyword_lists_per_launch_round'))" | tee -a gpurun_out/r3_callers.txt; }
mkdir -p gpurun_out
run --kw-threads 128
run --kw-threads 160
run --kw-threads 192 --kw-slots 384
run --kw-threads 256 --kw-slots 256
run --kw-threads 384 --kw-slots 192
