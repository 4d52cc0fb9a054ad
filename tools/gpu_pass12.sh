#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x --tb=short -p no:cacheprovider > gpurun_out/p12_tests.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/p12_tests.log
timeout 900 python bench.py > gpurun_out/bench_c4.json 2> gpurun_out/bench_c4.err; echo "bench rc=$?"; cut -c1-1500 gpurun_out/bench_c4.json; tail -3 gpurun_out/bench_c4.err
