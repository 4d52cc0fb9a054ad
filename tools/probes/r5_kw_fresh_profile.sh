# Round 5: where the host CPU of the keyword leg goes on a stream of FRESH queries (bench.py's stream since this round)
set -x
mkdir -p gpurun_out
RB_PROFILE_PER_THREAD=1 KW_PROFILE=gpurun_out/r5_kw_fresh.prof MSI_SEARCH_CPU_PROFILE=1 timeout 600 python tools/kw_leg.py --queries 3072 --fresh 4608 2>&1 | grep -v amdgpu.ids | tail -2
python tools/r3_symbolize.py gpurun_out/r5_kw_fresh.prof 60 > gpurun_out/r5_kw_fresh_profile.txt 2>&1
head -75 gpurun_out/r5_kw_fresh_profile.txt | cut -c1-200
rm -f gpurun_out/r5_kw_fresh.prof
