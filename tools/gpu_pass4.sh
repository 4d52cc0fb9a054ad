#!/bin/bash
# one-off: the group API tests and the concurrent distinct/sort/geo test, repeated, with full failure output
mkdir -p gpurun_out
python -m pytest tests/test_zz_group_gpu.py -m gpu -q --tb=short > gpurun_out/p4_group.log 2>&1; echo "group rc=$?"
for i in 1 2 3; do
  python -m pytest tests/test_zzz_distinct_gpu.py -m gpu -q --tb=short -k concurrent > gpurun_out/p4_conc_$i.log 2>&1; echo "conc $i rc=$?"
done
MSI_SEARCH_VM=0 python -m pytest tests/test_zzz_distinct_gpu.py -m gpu -q --tb=short -k concurrent > gpurun_out/p4_conc_direct.log 2>&1; echo "conc direct rc=$?"
tail -5 gpurun_out/p4_group.log; for i in 1 2 3; do tail -3 gpurun_out/p4_conc_$i.log; done; tail -3 gpurun_out/p4_conc_direct.log
