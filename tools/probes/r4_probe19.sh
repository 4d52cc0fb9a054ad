export GPU_MAX_HW_QUEUES=16
timeout 400 python bench.py --config c4 --no-rank --steps 8 --warmup 2 --no-pmc --no-cpu-baseline --no-also 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('c4 vector only', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d.get('legs',{}).get('scan_kernel_alone'))"
timeout 200 python bench.py --config c2 --steps 20 --warmup 3 --no-pmc --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('c2', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])"
