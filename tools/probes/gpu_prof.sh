#!/bin/bash
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bench -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pmc > $R/gpurun_out/prof_bench.log 2>&1; echo "rocprof rc=$?"
cd $R; f=$(find gpurun_out/prof_bench -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/bench_c4_kernel_stats.csv; head -6 gpurun_out/bench_c4_kernel_stats.csv | cut -c1-170
rm -rf gpurun_out/prof_bench
timeout 300 python bench.py --config c2 > gpurun_out/bench_c2.json 2>/dev/null; python -c "
import json; d=json.loads(open('gpurun_out/bench_c2.json').read().strip().splitlines()[-1]); print('c2', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d['parity']['mismatches'], d['config'].get('queries_per_hbm_sweep'))"
