# round 3: the keyword leg after the per-search arena (host allocations 6 700 -> ~800 per query)
mkdir -p gpurun_out
export RB_DETAILED=1 RB_DISTINCT_QUERIES=3072 GPU_MAX_HW_QUEUES=16
timeout 500 tools/bin/ranked_bench 10000000 200000 3 32 1 64 128 160 2>&1 | tee gpurun_out/r3_ranked_10m_arena.jsonl | python -c "
import sys, json
for line in sys.stdin:
    try: d = json.loads(line)
    except Exception: print(line.rstrip()[:300]); continue
    print(d['threads'], 'callers', d['queries_per_s'], 'q/s p50', d['p50_ms'], 'p99', d['p99_ms'], 'cpus', d['cpu'], 'vm', d['vm']['lists'] / max(1, d['vm']['rounds']))
"
RB_PROFILE=gpurun_out/r3_ranked_arena_profile.txt timeout 300 tools/bin/ranked_bench 10000000 200000 3 48 128 2>&1 | cut -c1-200
timeout 400 python bench.py --no-pmc --no-cpu-baseline --no-also --steps 10 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('bench', d['value'], d['ms_per_step'], 'frac', d['roofline']['frac'], 'legs', d['legs'])"
