export GPU_MAX_HW_QUEUES=16
timeout 600 python -m pytest tests/test_vs_gpu.py tests/test_zz_group_gpu.py tests/test_configs_gpu.py -m gpu -q -x 2>&1 | tail -3
timeout 400 python bench.py --config c4 --no-rank --steps 8 --warmup 2 --no-pmc --no-cpu-baseline --no-also 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('c4 vector only, pipelined', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])"
MSI_VS_PIPELINE=0 timeout 400 python bench.py --config c4 --no-rank --steps 8 --warmup 2 --no-pmc --no-cpu-baseline --no-also 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('c4 vector only, one stream', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])"
