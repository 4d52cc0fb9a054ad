#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_zzz_filter_gpu.py -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -3
for a in "--queries 768 --kw-threads 64" "--queries 3072 --kw-threads 64" "--queries 768 --kw-threads 24"; do
  echo "== $a"; timeout 600 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-pmc $a 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('legs'))"
done
