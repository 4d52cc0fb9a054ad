export GPU_MAX_HW_QUEUES=16
timeout 500 python tools/probes/r4_kw_classes.py 2>&1 | grep -v amdgpu.ids | tail -12
RB_DETAILED=1 timeout 300 tools/bin/ranked_bench 10000000 200000 3 64 128 160 160 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: continue
    print({k: d[k] for k in d if k in ('threads','queries_per_s','qps','p50_ms','cpus_used')} or str(d)[:200])
"
