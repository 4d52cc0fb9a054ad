# Round 6, twenty-third device call: the default command with 384 callers on pools of 448 slots (a step is 768 searches: two
# generations of 384 instead of three of 256; the twenty-second call's five-step run of that shape: 60.95 ms per step
# against 64.1-64.6) — the WHOLE default command, with the free HBM at its fullest moments in the line (config.hbm_free_gb_min)
set -x
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
export TMPDIR=/tmp
ulimit -c 0
cd $R
( time MSI_BENCH_CALLERS_PER_CPU=24 timeout 900 python bench.py --kw-threads 384 --kw-slots 448 2>gpurun_out/r6_bench_384x448.err | tail -1 > gpurun_out/r6_bench_384x448.json ) 2>&1 | tail -3
cut -c1-4200 gpurun_out/r6_bench_384x448.json
cp gpurun_out/bench_detail_c4_n1.json gpurun_out/r6_bench_384x448_detail.json
tail -3 gpurun_out/r6_bench_384x448.err | cut -c1-400
