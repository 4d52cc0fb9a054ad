#!/bin/bash
mkdir -p gpurun_out
export MSI_VS_SCAN_MATH=bf16x2
timeout 900 python -m pytest tests/test_vs_gpu.py tests/test_zzz_vs_update_gpu.py tests/test_configs_gpu.py tests/test_golden_fixtures.py tests/test_zz_group_gpu.py -m gpu -q -x --tb=short -p no:cacheprovider > gpurun_out/p34_tests.log 2>&1; echo "tests rc=$?"; grep -v "^$" gpurun_out/p34_tests.log | grep -iv "amdgpu.ids\|Librccl\|RCCL version\|HIP version\|ROCm version\|Hostname" | tail -4 | cut -c1-300
timeout 900 python bench.py --steps 6 --warmup 2 > gpurun_out/bench_c4_bf16x2.json 2> gpurun_out/bench_c4_bf16x2.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_c4_bf16x2.json').read().strip().splitlines()[-1])
print(d['value'], d['unit'], d['ms_per_step'], 'frac', d['roofline'].get('frac'), 'avg_ms', d['roofline'].get('avg_launch_ms'), 'traffic', d['roofline'].get('traffic'), 'parity', d.get('parity',{}).get('mismatches'), d.get('legs'), d['config'].get('queries_per_hbm_sweep'), d['config'].get('inexact_queries_last_step'))
PY
grep -v amdgpu.ids gpurun_out/bench_c4_bf16x2.err | tail -3 | cut -c1-300
