# First GPU call of the next round: what round 4 built after its GPU minutes were spent, on the device.
#  1. the corpus tests that so far only ran on the emulated kernels (phrases, word-prefix databases, synonyms)
#  2. the keyword leg by universe class with MSI_SEARCH_LATE_WAIT=0 / 1 (a task waits for its siblings and then moves its
#     bucket's sub-tree into the bucket's compact space: built and fuzzed in round 4, never timed)
set -x
export GPU_MAX_HW_QUEUES=16
mkdir -p gpurun_out
MSI_TEST_UNTRIED_ON_DEVICE=1 timeout 900 python -m pytest tests/test_configs_gpu.py -m gpu -q -x -k "phrases or word_prefix or synonyms or negative_terms" 2>&1 | grep -E "passed|failed|error" | tail -3
for W in 0 1; do
  echo "== MSI_SEARCH_LATE_WAIT=$W"
  MSI_SEARCH_LATE_WAIT=$W timeout 500 python tools/probes/r4_kw_classes.py 2>&1 | grep -v "amdgpu.ids\|first pass" | tail -9
done
