#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_zz_vm_gpu.py tests/test_search_gpu.py -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -3
timeout 900 python bench.py > gpurun_out/bench_c4.json 2> gpurun_out/bench_c4.err; echo "bench rc=$?"; python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_c4.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d.get('legs'), d['roofline']['frac'], d['roofline']['traffic'], d['parity']['mismatches'], d['parity'].get('keyword'))
PY
grep -v amdgpu.ids gpurun_out/bench_c4.err | tail -3
