# Round 5, last GPU call: the whole GPU tier on the final tree, smoke(), the default bench command as the driver runs it
set -x
mkdir -p gpurun_out
(time timeout 1500 python -m pytest tests -m gpu -x -q) 2>&1 | tail -6
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r5_bench_final.log 2> gpurun_out/r5_bench_final.err
tail -1 gpurun_out/r5_bench_final.log | cut -c1-4200
tail -1 gpurun_out/r5_bench_final.log | wc -c
