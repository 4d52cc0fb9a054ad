#!/bin/bash
# one-off: BQ at 10 M x 768, kernel timeline of the keyword search after the fence change
mkdir -p gpurun_out
timeout 600 python tools/bench_configs.py bq --rows 10000000 --dim 768 --reps 5 > gpurun_out/p7_bq_10m.jsonl 2> gpurun_out/p7_bq.err; echo rc=$?; cat gpurun_out/p7_bq_10m.jsonl
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_ranked10b -o ranked -- $R/tools/bin/ranked_bench 10000000 200000 3 16 32 > $R/gpurun_out/prof_ranked10b.log 2>&1; echo rc=$?
cd $R
grep qps -A0 gpurun_out/prof_ranked10b.log | cut -c1-300 | tail -2
sed 's/"config.*"queries_per_s"/"qps"/' gpurun_out/prof_ranked10b.log | grep '"qps"' | cut -c1-200
rm -rf gpurun_out/prof_ranked10b/*.db 2>/dev/null
