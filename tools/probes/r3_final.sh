# round-3 validation pass on the MI355X box: the GPU tier, smoke, the driver's own bench command, a kernel-trace summary
set -x
mkdir -p gpurun_out
t0=$(date +%s)
timeout 1100 python -m pytest tests -m gpu -q -x 2>&1 | tail -6
echo "tests took $(( $(date +%s) - t0 )) s"
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
t0=$(date +%s)
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r3_bench_c4_final.json 2> gpurun_out/r3_bench_c4_final.err; echo bench rc=$?
echo "bench took $(( $(date +%s) - t0 )) s"
cut -c1-1200 gpurun_out/r3_bench_c4_final.json; tail -3 gpurun_out/r3_bench_c4_final.err
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3_prof_bench -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-also --no-pmc > $GRAFT_REPO_ROOT/gpurun_out/r3_prof_bench.log 2>&1; echo rocprof rc=$?
cd $GRAFT_REPO_ROOT; f=$(find gpurun_out/r3_prof_bench -name "*kernel_stats.csv" | head -1); echo $f; head -14 "$f" | cut -c1-220
tail -1 gpurun_out/r3_prof_bench.log | cut -c1-400
