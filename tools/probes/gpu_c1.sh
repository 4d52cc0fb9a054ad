#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python bench.py --config c1 > gpurun_out/bench_c1.json 2> gpurun_out/bench_c1.err; echo "c1 rc=$?"; cut -c1-1800 gpurun_out/bench_c1.json; grep -v amdgpu.ids gpurun_out/bench_c1.err | tail -5 | cut -c1-300
