#!/bin/bash
python - <<'PY'
import os
print("affinity before imports", len(os.sched_getaffinity(0)))
import numpy
print("after numpy", len(os.sched_getaffinity(0)))
import torch
print("after torch", len(os.sched_getaffinity(0)), torch.get_num_threads())
PY
cat /sys/fs/cgroup/cpu.max
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pmc 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('legs'))"
echo "== OMP 1"; OMP_NUM_THREADS=1 MKL_NUM_THREADS=1 timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pmc 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('legs'))"
