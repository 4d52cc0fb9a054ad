export GPU_MAX_HW_QUEUES=16
for S in 8 0 16; do
MSI_VS_SPARE_CUS=$S timeout 400 python bench.py --config c4 --no-rank --steps 8 --warmup 2 --no-pmc --no-cpu-baseline --no-also 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('c4 vector only, spare CUs $S:', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])"
done
