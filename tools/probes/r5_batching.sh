# Round 5: the combiner's batching knobs at 256 callers (a list waits 199 us "for company" of its ~780 us per round)
for cfg in "32 200" "16 200" "32 100" "16 100"; do
  set -- $cfg
  echo "== MSI_VM_BATCH_CAP=$1 MSI_VM_BATCH_WAIT_US=$2"
  MSI_VM_BATCH_CAP=$1 MSI_VM_BATCH_WAIT_US=$2 timeout 300 python tools/kw_leg.py --callers 256 --queries 3072 --passes 3 2>&1 | grep -v amdgpu.ids | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('q/s', d['queries_per_s'], 'cpus', d['host_cpus_used'], d['vm'])"
done
