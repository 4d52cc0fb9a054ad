mkdir -p gpurun_out
export RB_DETAILED=1 RB_DISTINCT_QUERIES=512
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_ranked10c -o ranked -- $GRAFT_REPO_ROOT/tools/bin/ranked_bench 10000000 200000 3 24 1 64 > $GRAFT_REPO_ROOT/gpurun_out/prof_ranked10c.log 2>&1; echo rc=$?
cd $GRAFT_REPO_ROOT
cat gpurun_out/prof_ranked10c.log | grep queries_per_s | sed 's/"config.*"queries_per_s"/"qps"/' | cut -c1-200
grep -o '"compact_space.*' gpurun_out/prof_ranked10c.log | head -3
python - <<'PY'
import csv,collections,statistics,glob
import numpy as np
f=glob.glob('gpurun_out/prof_ranked10c/**/ranked_kernel_trace.csv',recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if 'vm_kernel' in r['Kernel_Name']]
print(len(rows),'vm launches')
gx=np.array([int(r['Grid_Size_X'])//256 for r in rows]); gy=np.array([int(r['Grid_Size_Y']) for r in rows])
d=np.array([(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in rows])
st=np.array([int(r['Start_Timestamp']) for r in rows]); en=np.array([int(r['End_Timestamp']) for r in rows])
for name,m in (('wide (>=100 chunks)',gx>=100),('narrow (<100 chunks)',gx<100)):
    for lo,hi in ((1,1),(2,7),(8,15),(16,64)):
        mm=m&(gy>=lo)&(gy<=hi)
        if mm.sum(): print(name,'lists',lo,hi,'n',mm.sum(),'p50 us',round(float(np.median(d[mm])),1),'p90',round(float(np.percentile(d[mm],90)),1),'sum ms',round(float(d[mm].sum()/1e3),1))
# last window with multi-list launches
t0=st[gy>=4].min() if (gy>=4).any() else st.min()
w=st>=t0
print('64-caller window ms',(en[w].max()-t0)/1e6,'kernel-time/wall',d[w].sum()*1e3/(en[w].max()-t0))
PY
