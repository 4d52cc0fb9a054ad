# Round 6, twenty-fifth device call: the whole GPU tier and smoke() on the final tree (the library with the checked rank-table
# allocation, bench.py with its rows freed: its default command ran in the twenty-fourth call)
set -x
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
export TMPDIR=/tmp
ulimit -c 0
cd $R
( time timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 ) > gpurun_out/r6_gpu_tier_final.log 2>&1
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -a -v amdgpu.ids | tail -1 >> gpurun_out/r6_gpu_tier_final.log
cat gpurun_out/r6_gpu_tier_final.log
