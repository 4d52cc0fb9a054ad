# Round 5, last GPU call of the round: the command-list / ranked-search tests on the device with the round's last changes
# (unmaterialised universe, no rank tables for a universe known to be too large, reaper off), then the default bench command
# exactly as the driver runs it
set -x
mkdir -p gpurun_out
(time timeout 170 python -m pytest tests/test_zz_vm_gpu.py tests/test_zz_levels_per_wait_gpu.py tests/test_search_gpu.py -m gpu -x -q) 2>&1 | tail -12
timeout 330 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r5_bench_final2.log 2> gpurun_out/r5_bench_final2.err
echo "bench rc $?"
tail -1 gpurun_out/r5_bench_final2.log | cut -c1-4200
tail -1 gpurun_out/r5_bench_final2.log | wc -c
tail -3 gpurun_out/r5_bench_final2.err | cut -c1-300
