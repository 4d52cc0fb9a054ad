#!/bin/bash
python - <<'PY'
import ctypes, os
import torch, meilisearch_amd as ma
ma._lib.lib()
for l in open('/proc/self/maps'):
    if 'libamdhip64' in l or 'libhsa-runtime' in l:
        print(l.split()[-1]); 
PY
echo "== preload system HIP runtime"
LD_PRELOAD=/opt/rocm/lib/libamdhip64.so timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pmc 2>&1 | tail -3 | python -c "
import json,sys
t=sys.stdin.read().strip().splitlines()
try:
    d=json.loads(t[-1]); print(d['value'], d['ms_per_step'], d.get('legs'))
except Exception as e: print('ERR', t[-3:])"
