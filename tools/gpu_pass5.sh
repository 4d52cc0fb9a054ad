#!/bin/bash
# one-off: (1) try to reproduce the flaky concurrent distinct/sort/geo failure in suite order, (2) kernel timeline of the
# ranked keyword search under load (2 M and 10 M documents)
mkdir -p gpurun_out
for i in 1 2 3; do
  timeout 600 python -m pytest tests/test_zz_rules_gpu.py tests/test_zz_vm_gpu.py tests/test_zzz_distinct_gpu.py -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/p5_suite_$i.log 2>&1; echo "suite $i rc=$?"
  tail -2 gpurun_out/p5_suite_$i.log
done
for i in 1 2 3 4 5 6; do
  timeout 300 python -m pytest tests/test_zzz_distinct_gpu.py -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/p5_file_$i.log 2>&1; echo "file $i rc=$?"
done
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_ranked -o ranked -- $R/tools/bin/ranked_bench 2000000 200000 3 24 64 > $R/gpurun_out/prof_ranked.log 2>&1; echo rc=$?
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_ranked10 -o ranked -- $R/tools/bin/ranked_bench 10000000 200000 3 16 32 > $R/gpurun_out/prof_ranked10.log 2>&1; echo rc=$?
cd $R
for d in prof_ranked prof_ranked10; do
grep qps -A0 gpurun_out/$d.log | cut -c1-300 | tail -2
python - $d <<'PY'
import csv,collections,statistics,sys
d=sys.argv[1]
rows=list(csv.DictReader(open('gpurun_out/%s/ranked_kernel_trace.csv'%d)))
rows=[r for r in rows if 'vm_kernel' in r['Kernel_Name'] or 'copyBuffer' in r['Kernel_Name']]
vm=[r for r in rows if 'vm_kernel' in r['Kernel_Name']]
print(d, len(vm), 'vm launches; columns', list(rows[0].keys()))
t0=min(int(r['Start_Timestamp']) for r in vm); t1=max(int(r['End_Timestamp']) for r in vm)
# steady-state window: the last 60 % of the run
lo=t0+(t1-t0)*0.4
vm=[r for r in vm if int(r['Start_Timestamp'])>=lo]
by=collections.defaultdict(list)
for r in vm:
    gy=int(r.get('Grid_Size_Y',256))//1
    by[gy].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for k in sorted(by):
    dd=sorted(by[k]); print('grid_y=%d: n=%d p50=%.1f p90=%.1f max=%.1f us'%(k,len(dd),statistics.median(dd),dd[int(len(dd)*0.9)],dd[-1]))
ev=[]
for r in vm:
    ev.append((int(r['Start_Timestamp']),1)); ev.append((int(r['End_Timestamp']),-1))
ev.sort(); cur=0; last=ev[0][0]; acc=collections.Counter()
for t,dlt in ev:
    acc[cur]+=t-last; last=t; cur+=dlt
tot=sum(acc.values()); print('vm kernels in flight (fraction of time):', {k: round(v/tot,3) for k,v in sorted(acc.items())})
print('launch rate: %.1f vm kernels per ms'%(len(vm)/((t1-lo)/1e6)))
# per queue: gap between the end of a kernel and the start of the next kernel of the same queue
byq=collections.defaultdict(list)
for r in rows:
    if int(r['Start_Timestamp'])>=lo: byq[r.get('Queue_Id','?')].append((int(r['Start_Timestamp']),int(r['End_Timestamp']),r['Kernel_Name'][:12]))
gaps=[]
for q,l in byq.items():
    l.sort()
    for a,b in zip(l,l[1:]):
        gaps.append((b[0]-a[1])/1e3)
gaps.sort()
print('queues',len(byq),'gap between consecutive kernels of a queue: p10=%.1f p50=%.1f p90=%.1f us'%(gaps[len(gaps)//10],gaps[len(gaps)//2],gaps[len(gaps)*9//10]))
PY
done
head -c 20000000 gpurun_out/prof_ranked10/ranked_kernel_trace.csv > gpurun_out/ranked10_kernel_trace_head.csv
rm -rf gpurun_out/prof_ranked/*.db gpurun_out/prof_ranked10/*.db 2>/dev/null
