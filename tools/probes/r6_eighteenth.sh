# Round 6, eighteenth device call: what the driver runs at round end, on this tree — the whole GPU tier, smoke(), the default
# bench command (now with legs.hybrid_legs_side_by_side) — and the rocprofv3 kernel trace of the C4 step of the same command,
# whose average sweep duration the line's roofline object must agree with (profiles/r6_bench_c4_kernel_stats.csv)
set -x
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
export TMPDIR=/tmp
cd $R
( time timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 ) > gpurun_out/r6_gpu_tier_final.log 2>&1
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -a -v amdgpu.ids | tail -1 >> gpurun_out/r6_gpu_tier_final.log
cat gpurun_out/r6_gpu_tier_final.log
( time timeout 900 python bench.py 2>gpurun_out/r6_bench_default_final.err | tail -1 > gpurun_out/r6_bench_default_final.json ) 2>&1 | tail -3
cut -c1-3000 gpurun_out/r6_bench_default_final.json
tail -5 gpurun_out/r6_bench_default_final.err
cd /tmp
rm -rf /tmp/tr_c4
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr_c4 -o tr -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-also --no-pmc --kw-features 0 --no-overlapped-leg > /tmp/tr_c4.log 2>&1
F=$(find /tmp/tr_c4 -name "*kernel_stats.csv" | head -1)
[ -n "$F" ] && cp $F $R/gpurun_out/r6_bench_c4_kernel_stats.csv && head -14 $F | cut -c1-220
tail -1 /tmp/tr_c4.log | cut -c1-1200 | tee $R/gpurun_out/r6_bench_c4_traced_line.json
