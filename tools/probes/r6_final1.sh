# Round 6: the whole GPU tier, then the default command (the line the driver will take), then serial vs fifo order of a step's searches
set -x
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
( time timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | grep -a "passed\|failed\|error" | tail -3 ) 2>&1 | tee gpurun_out/r6_gpu_tier.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee -a gpurun_out/r6_gpu_tier.log
( time timeout 900 python bench.py > gpurun_out/r6_bench_default2.log 2>gpurun_out/r6_bench_default2.err ) 2>&1 | tail -3
tail -1 gpurun_out/r6_bench_default2.log | cut -c1-4200
RB_ORDER=fifo timeout 500 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-also --no-pmc --kw-features 0 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.readline())
print('fifo order: value', d['value'], 'ms_per_step', d['ms_per_step'], 'legs', d['legs'])" | tee gpurun_out/r6_order_fifo.log
