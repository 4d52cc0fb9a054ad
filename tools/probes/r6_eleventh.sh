# Round 6, eleventh device call: who notices the rounds' completions now that the host has CPUs to spare — the reaper thread
# (MSI_VM_REAPER, per round: one process, same index and cache) and two combiners
set -x
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
MSI_VM_REAPER=0 MSI_SEARCH_CPU_PROFILE=1 timeout 900 python tools/kw_leg.py --callers 256 --queries 3072 --segment 3072 --sweep "0:256,1:256,0:256,1:256" 2>&1 | grep -a -v amdgpu.ids | tail -4 | tee gpurun_out/r6_reaper.log | cut -c1-500
for c in 1 2; do
  MSI_VM_COMBINERS=$c MSI_SEARCH_CPU_PROFILE=1 timeout 900 python tools/kw_leg.py --callers 256 --queries 3072 --fresh 4608 2>&1 | grep -a -v amdgpu.ids | tail -1 | sed "s/^/combiners=$c /"
done | tee gpurun_out/r6_combiners.log | cut -c1-600
for p in 20 5 0; do
  MSI_VM_POLL_SLEEP_US=$p MSI_SEARCH_CPU_PROFILE=1 timeout 900 python tools/kw_leg.py --callers 256 --queries 3072 --fresh 4608 2>&1 | grep -a -v amdgpu.ids | tail -1 | sed "s/^/poll_sleep_us=$p /"
done | tee gpurun_out/r6_poll.log | cut -c1-600
