set -x
mkdir -p gpurun_out
timeout 400 python -m pytest tests -m gpu -q 2>&1 | tail -4
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 400 python bench.py > gpurun_out/bench_c4.json 2> gpurun_out/bench_c4.err; echo bench rc=$?; cat gpurun_out/bench_c4.json | cut -c1-900; tail -2 gpurun_out/bench_c4.err
cd /tmp && export TMPDIR=/tmp
timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_bench -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_bench.log 2>&1; echo rocprof rc=$?
cd $GRAFT_REPO_ROOT; f=$(find gpurun_out/prof_bench -name "*kernel_stats.csv" | head -1); echo $f; head -12 "$f" | cut -c1-200
# rows of SURVEY 8 f2 / f3 that were written without GPU budget (round 1): per-bucket cost of Sort / distinct / GeoSort and
# the incremental store update, then their kernels' rocprof summary
cd $GRAFT_REPO_ROOT
timeout 300 python tools/bench_configs.py rules --rows 2000000 --reps 7 > gpurun_out/rules_2m.jsonl 2> gpurun_out/rules_2m.err; echo rules rc=$?; cat gpurun_out/rules_2m.jsonl | cut -c1-300
timeout 300 python tools/bench_configs.py update --rows 1000000 --dim 384 --reps 5 > gpurun_out/update_c2.jsonl 2> gpurun_out/update_c2.err; echo update rc=$?; cat gpurun_out/update_c2.jsonl | cut -c1-400
cd /tmp && export TMPDIR=/tmp
timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_rules -o rules -- python $GRAFT_REPO_ROOT/tools/bench_configs.py rules --rows 2000000 --reps 3 > $GRAFT_REPO_ROOT/gpurun_out/prof_rules.log 2>&1; echo rocprof rules rc=$?
cd $GRAFT_REPO_ROOT
# C3 alone (dict_match not co-scheduled with the scan): words/s at the BASELINE batch sizes, CPU sample 4096 words
timeout 400 python tools/bench_configs.py c3 --batches 1 64 1024 8192 --cpu-sample 4096 --reps 5 > gpurun_out/c3_sweep.jsonl 2> gpurun_out/c3_sweep.err; echo c3 rc=$?; cat gpurun_out/c3_sweep.jsonl | cut -c1-500
nproc; lscpu | grep -i "model name"
