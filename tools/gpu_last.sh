#!/bin/bash
# one short call: the vm/tasks parity tests, then the detailed-score ranked bench at 1 and 64 threads
timeout 40 python -m pytest tests/test_zz_vm_gpu.py -q -x -m gpu 2>&1 | tail -1
RB_DETAILED=1 timeout 40 tools/bin/ranked_bench 10000000 200000 3 32 1 64 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    print(d['threads'], d['queries_per_s'], 'p50', d['p50_ms'], 'wait_us', d['device_wait_us_per_query'], 'cb_us', d['callback_us_per_query'], d['cpu'])" | tee gpurun_out/r2_last_rb.txt
