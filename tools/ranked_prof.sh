mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_ranked -o ranked -- $GRAFT_REPO_ROOT/tools/bin/ranked_bench 2000000 200000 3 16 1 64 > $GRAFT_REPO_ROOT/gpurun_out/prof_ranked.log 2>&1; echo rc=$?
cd $GRAFT_REPO_ROOT; cat gpurun_out/prof_ranked.log | cut -c1-300 | tail -4
head -8 gpurun_out/prof_ranked/ranked_kernel_stats.csv | cut -c1-220
python - <<'PY'
import csv,collections
rows=list(csv.DictReader(open('gpurun_out/prof_ranked/ranked_kernel_trace.csv')))
vm=[r for r in rows if 'vm_kernel' in r['Kernel_Name']]
print(len(vm), 'vm launches')
import statistics
d=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in vm]
print('dur us p50/p90/max', statistics.median(d), sorted(d)[int(len(d)*0.9)], max(d))
g=collections.Counter((r['Grid_Size_X'] if 'Grid_Size_X' in r else r.get('Grid_Size'), r.get('Grid_Size_Y')) for r in vm)
print(g.most_common(8))
st=sorted(int(r['Start_Timestamp']) for r in vm)
gaps=[(b-a)/1e3 for a,b in zip(st,st[1:])]
print('start-to-start us p50/p90', statistics.median(gaps), sorted(gaps)[int(len(gaps)*0.9)])
PY
