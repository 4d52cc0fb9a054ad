#!/bin/bash
# round-end measurement pass: every bench.py config, the kernel summary of the headline line, the keyword-search sweeps
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
for c in c4 c2 c3 c5; do
  timeout 900 python bench.py --config $c > gpurun_out/bench_$c.json 2> gpurun_out/bench_$c.err; echo "$c rc=$?"
  python - $c <<'PY'
import json,sys
c=sys.argv[1]
try:
    d=json.loads(open(f'gpurun_out/bench_{c}.json').read().strip().splitlines()[-1])
    print(c, d['value'], d['unit'], d['ms_per_step'], 'frac', d['roofline'].get('frac'), 'traffic', d['roofline'].get('traffic'), 'cpu', d.get('cpu_baseline',{}).get('value'), 'parity', d.get('parity',{}).get('mismatches'), d.get('legs'))
except Exception as e: print(c, 'ERR', e)
PY
done
timeout 300 tools/bin/ranked_bench 2000000 200000 3 32 1 16 64 > gpurun_out/ranked_2m_plain.jsonl 2>/dev/null
RB_DETAILED=1 timeout 300 tools/bin/ranked_bench 2000000 200000 3 32 1 16 64 > gpurun_out/ranked_2m_detailed.jsonl 2>/dev/null
timeout 400 tools/bin/ranked_bench 10000000 200000 3 32 1 16 64 > gpurun_out/ranked_10m_plain.jsonl 2>/dev/null
RB_DETAILED=1 timeout 400 tools/bin/ranked_bench 10000000 200000 3 32 1 16 64 > gpurun_out/ranked_10m_detailed.jsonl 2>/dev/null
for f in ranked_2m_plain ranked_2m_detailed ranked_10m_plain ranked_10m_detailed; do echo $f; sed 's/"config.*"queries_per_s"/"qps"/' gpurun_out/$f.jsonl | cut -c1-110; done
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bench -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pmc > $R/gpurun_out/prof_bench.log 2>&1; echo "rocprof rc=$?"
cd $R; f=$(find gpurun_out/prof_bench -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/bench_c4_kernel_stats.csv; head -8 gpurun_out/bench_c4_kernel_stats.csv | cut -c1-160
rm -rf gpurun_out/prof_bench
