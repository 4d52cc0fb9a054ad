#!/bin/bash
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q -x --tb=short -p no:cacheprovider > gpurun_out/final_tests.log 2>&1; echo "tests rc=$?"; grep -v "^$" gpurun_out/final_tests.log | grep -iv "amdgpu.ids\|Librccl\|RCCL version\|HIP version\|ROCm version\|Hostname" | tail -3 | cut -c1-200
timeout 900 python bench.py > gpurun_out/bench_c4.json 2> gpurun_out/bench_c4.err; echo "c4 rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_c4.json').read().strip().splitlines()[-1])
print(d['value'], d['unit'], d['ms_per_step'], 'frac', d['roofline'].get('frac'), 'traffic', d['roofline'].get('traffic'), 'cpu', d.get('cpu_baseline',{}).get('value'), 'parity', d.get('parity',{}).get('mismatches'), d.get('legs'))
PY
timeout 400 tools/bin/ranked_bench 10000000 200000 3 32 1 16 64 > gpurun_out/ranked_10m_plain.jsonl 2>/dev/null
RB_DETAILED=1 timeout 400 tools/bin/ranked_bench 10000000 200000 3 32 1 16 64 > gpurun_out/ranked_10m_detailed.jsonl 2>/dev/null
timeout 300 tools/bin/ranked_bench 2000000 200000 3 32 1 16 64 > gpurun_out/ranked_2m_plain.jsonl 2>/dev/null
RB_DETAILED=1 timeout 300 tools/bin/ranked_bench 2000000 200000 3 32 1 16 64 > gpurun_out/ranked_2m_detailed.jsonl 2>/dev/null
for f in ranked_2m_plain ranked_2m_detailed ranked_10m_plain ranked_10m_detailed; do echo $f; sed 's/"config.*"queries_per_s"/"qps"/' gpurun_out/$f.jsonl | cut -c1-60; done
