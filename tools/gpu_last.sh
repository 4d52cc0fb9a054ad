#!/bin/bash
# one short call: cgroup quota, then the detailed-score ranked bench at 64 and 16 threads
(cat /sys/fs/cgroup/cpu.max; nproc) 2>&1 | tr '\n' ' ' | tee gpurun_out/r2_last_rb2.txt; echo | tee -a gpurun_out/r2_last_rb2.txt
RB_DETAILED=1 timeout 40 tools/bin/ranked_bench 10000000 200000 3 32 64 16 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    print(d['threads'], d['queries_per_s'], 'p50', d['p50_ms'], 'wait_us', d['device_wait_us_per_query'], 'cb_us', d['callback_us_per_query'], d['cpu'])" | tee -a gpurun_out/r2_last_rb2.txt
