# round 3: where the keyword leg's time goes at 10 M documents (detailed scores, 3 terms)
mkdir -p gpurun_out
export RB_DETAILED=1 RB_DISTINCT_QUERIES=3072
echo "== throughput by caller threads"
timeout 400 tools/bin/ranked_bench 10000000 200000 3 24 1 16 64 128 2>&1 | tee gpurun_out/r3_ranked_10m_probe.jsonl | sed 's/"config.*"queries_per_s"/"qps"/' | cut -c1-1500
echo "== kernel trace, 64 callers"
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_ranked10 -o ranked -- $GRAFT_REPO_ROOT/tools/bin/ranked_bench 10000000 200000 3 24 64 > $GRAFT_REPO_ROOT/gpurun_out/prof_ranked10.log 2>&1; echo rc=$?
cd $GRAFT_REPO_ROOT
head -8 gpurun_out/prof_ranked10/ranked_kernel_stats.csv | cut -c1-220
python - <<'PY'
import csv,collections,statistics,glob
f=glob.glob('gpurun_out/prof_ranked10/**/ranked_kernel_trace.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
vm=[r for r in rows if 'vm_kernel' in r['Kernel_Name']]
print(len(vm), 'vm launches')
by=collections.defaultdict(list)
for r in vm:
    gy=int(r.get('Grid_Size_Y',1))
    by[min(gy//4*4,64)].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for k in sorted(by):
    d=sorted(by[k]); print('lists>=%d: n=%d p50=%.1f p90=%.1f max=%.1f us'%(k,len(d),statistics.median(d),d[int(len(d)*0.9)],d[-1]))
ev=[]
for r in vm:
    ev.append((int(r['Start_Timestamp']),1)); ev.append((int(r['End_Timestamp']),-1))
ev.sort(); cur=0; last=ev[0][0]; acc=collections.Counter()
for t,dlt in ev:
    acc[cur]+=t-last; last=t; cur+=dlt
tot=sum(acc.values()); print('vm kernels in flight (fraction of time):', {k: round(v/tot,3) for k,v in sorted(acc.items())})
PY
