# round 3: what the combiner's spinning buys on one GPU (it costs a CPU per rank, which matters with 8 ranks on one host)
run() { echo "== $*"; env "$@" python bench.py --no-pmc --no-cpu-baseline --no-also --steps 8 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['value'], 'q/s', d['ms_per_step'], 'ms/step; kw_only', d['legs'].get('keyword_only_queries_per_s'), 'cpus', d['legs'].get('keyword_only_host_cpus_used'), 'lists/round', d['legs'].get('keyword_lists_per_launch_round'))" | tee -a gpurun_out/r3_pollsleep.txt; }
mkdir -p gpurun_out
run X=1
run MSI_VM_POLL_SLEEP_US=10
run MSI_VM_POLL_SLEEP_US=30
