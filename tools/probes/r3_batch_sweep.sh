# round 3: combiner batching knobs and caller counts after the host-side cuts (arena, futex micro-batcher)
mkdir -p gpurun_out
export RB_DETAILED=1 RB_DISTINCT_QUERIES=3072 GPU_MAX_HW_QUEUES=16
run() {  # label, callers, env...
  label=$1; callers=$2; shift 2
  env "$@" timeout 120 tools/bin/ranked_bench 10000000 200000 3 64 $callers 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    try: d = json.loads(line)
    except Exception: continue
    print('$label', d['threads'], 'callers', d['queries_per_s'], 'q/s p50', d['p50_ms'], 'p99', d['p99_ms'], 'cpus', d['cpu']['cpus_used'], 'lists/round', round(d['vm']['lists'] / max(1, d['vm']['rounds']), 1), 'packed', d['vm']['us_packed_per_list'], 'after', d['vm']['us_after_launch_per_list'])
" | tee -a gpurun_out/r3_batch_sweep.txt
}
run default 128 X=1
run cap64 128 MSI_VM_BATCH_CAP=64
run wait100 128 MSI_VM_BATCH_WAIT_US=100
run wait400cap64 128 MSI_VM_BATCH_WAIT_US=400 MSI_VM_BATCH_CAP=64
run default 160 X=1
run cap64 160 MSI_VM_BATCH_CAP=64
run default 128 X=1
