# Round 5: the two legs of the C4 step side by side now that the vector leg is the int8 sweep (98 KB of LDS per workgroup, not
# 147: a command-list workgroup fits beside it when the sweep's registers leave room — kernel shape 3: 182 VGPRs)
mkdir -p gpurun_out
for cfg in "serial 0" "overlap 0" "overlap 3" "serial 3"; do
  set -- $cfg
  echo "== legs $1, MSI_VS_I8_VARIANT=$2"
  MSI_VS_I8_VARIANT=$2 timeout 400 python bench.py --legs $1 --steps 10 --warmup 3 --no-cpu-baseline --no-also --no-pmc 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('value', d['value'], 'ms_per_step', d['ms_per_step'], 'p50', d['p50_latency_ms'], 'roofline frac', d['roofline']['frac'], 'sweep ms', d['roofline']['avg_launch_ms'], 'legs', d.get('legs'))"
done
