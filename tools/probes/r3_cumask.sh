# round 3: the scan on a subset of the CUs, the keyword searches beside it (bench.py --legs overlap)
run() { echo "== $*"; env "$@" python bench.py --no-pmc --no-cpu-baseline --no-also --steps 8 --warmup 3 $LEGS 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['value'], 'q/s', d['ms_per_step'], 'ms/step; scan frac in step', d['roofline']['frac'], d['roofline']['avg_launch_ms'], 'ms; kw_only', d['legs'].get('keyword_only_queries_per_s'), 'vec_only', d['legs'].get('vector_only_queries_per_s'), 'parity', d.get('parity',{}).get('mismatches'))" | tee -a gpurun_out/r3_cumask.txt; }
mkdir -p gpurun_out
LEGS="--legs overlap" run MSI_SCAN_CUS=224
LEGS="--legs overlap" run MSI_SCAN_CUS=192
LEGS="--legs overlap" run MSI_SCAN_CUS=160
LEGS="--legs serial" run X=1
