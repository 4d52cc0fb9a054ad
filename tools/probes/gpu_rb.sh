#!/bin/bash
RB_DETAILED=1 timeout 400 tools/bin/ranked_bench 10000000 200000 3 32 1 64 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    print(d['threads'], d['queries_per_s'], 'p50', d['p50_ms'], 'wait_us', d['device_wait_us_per_query'], 'cb_us', d['callback_us_per_query'], d['cpu'])"
