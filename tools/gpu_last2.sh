#!/bin/bash
for a in "" "--serial-legs"; do
timeout 600 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-pmc $a 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$a', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('avg_launch_ms'))"
done
