#!/bin/bash
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pmc 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['roofline']['frac'], d['legs'])"
