mkdir -p gpurun_out
timeout 900 python bench.py > gpurun_out/bench_c4.json 2> gpurun_out/bench_c4.err; echo "c4 rc=$?"; cut -c1-4000 gpurun_out/bench_c4.json; tail -5 gpurun_out/bench_c4.err
