#!/bin/bash
mkdir -p gpurun_out
for t in 64 128; do
timeout 900 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-pmc --kw-threads $t 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('kw-threads', $t, d['value'], d['ms_per_step'], d.get('legs'))"
done
