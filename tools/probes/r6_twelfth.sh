# Round 6, twelfth device call: a round takes ANY free arena of its class (not the next in rotation), and more streams / hardware
# queues for the rounds — each configuration its own process on one box
set -x
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
timeout 600 python -m pytest -x -q -m gpu tests/test_zz_vm_gpu.py tests/test_search_gpu.py 2>&1 | grep -a "passed\|failed\|error" | tail -2 | tee gpurun_out/r6_twelfth_tests.log
run() {
  label="$1"; shift
  env "$@" MSI_SEARCH_CPU_PROFILE=1 timeout 900 python tools/kw_leg.py --callers 256 --queries 3072 --fresh 4608 2>&1 | grep -a -v amdgpu.ids | tail -1 | sed "s/^/$label /"
}
{
  run "any_free_arena,16_streams" MSI_VM_STREAMS=16
  run "fifo_arena,16_streams" MSI_VM_ARENA_FIFO=1
  run "any_free_arena,32_streams,16_hw_queues" MSI_VM_STREAMS=32
  run "any_free_arena,32_streams,32_hw_queues" MSI_VM_STREAMS=32 GPU_MAX_HW_QUEUES=32
  run "any_free_arena,24_streams,24_hw_queues" MSI_VM_STREAMS=24 GPU_MAX_HW_QUEUES=24
  run "any_free_arena,16_streams,again" MSI_VM_STREAMS=16
} | tee gpurun_out/r6_arenas.log | cut -c1-600
