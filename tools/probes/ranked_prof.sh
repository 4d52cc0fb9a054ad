mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_ranked -o ranked -- $GRAFT_REPO_ROOT/tools/bin/ranked_bench 2000000 200000 3 24 64 > $GRAFT_REPO_ROOT/gpurun_out/prof_ranked.log 2>&1; echo rc=$?
cd $GRAFT_REPO_ROOT; cat gpurun_out/prof_ranked.log | grep qps -A0 | cut -c1-300 | tail -4
head -6 gpurun_out/prof_ranked/ranked_kernel_stats.csv | cut -c1-200
python - <<'PY'
import csv,collections,statistics
rows=list(csv.DictReader(open('gpurun_out/prof_ranked/ranked_kernel_trace.csv')))
vm=[r for r in rows if 'vm_kernel' in r['Kernel_Name']]
print(len(vm), 'vm launches; columns', list(rows[0].keys())[:20])
by=collections.defaultdict(list)
for r in vm:
    gy=int(r.get('Grid_Size_Y',1))
    by[min(gy//8*8,64)].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for k in sorted(by):
    d=sorted(by[k]); print('lists>=%d: n=%d p50=%.1f p90=%.1f max=%.1f us'%(k,len(d),statistics.median(d),d[int(len(d)*0.9)],d[-1]))
# overlap: how many vm kernels run concurrently on average
ev=[]
for r in vm:
    ev.append((int(r['Start_Timestamp']),1)); ev.append((int(r['End_Timestamp']),-1))
ev.sort(); cur=0; last=ev[0][0]; acc=collections.Counter()
for t,dlt in ev:
    acc[cur]+=t-last; last=t; cur+=dlt
tot=sum(acc.values()); print('concurrency histogram (fraction of time):', {k: round(v/tot,3) for k,v in sorted(acc.items())})
PY
