# round 3: HBM bytes of the keyword leg's vm_kernel at 10 M documents (separate --pmc passes, no tracing domains)
mkdir -p gpurun_out
export RB_DETAILED=1 RB_DISTINCT_QUERIES=3072
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 400 rocprofv3 --pmc $C --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_ranked10_$C -o pmc -- $GRAFT_REPO_ROOT/tools/bin/ranked_bench 10000000 200000 3 24 64 > $GRAFT_REPO_ROOT/gpurun_out/pmc_ranked10_$C.log 2>&1; echo $C rc=$?
done
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv,glob,json
out={}
for C in ("FETCH_SIZE","WRITE_SIZE"):
    f=glob.glob('gpurun_out/pmc_ranked10_%s/**/*counter_collection.csv'%C,recursive=True)[0]
    rows=[r for r in csv.DictReader(open(f)) if 'vm_kernel' in r['Kernel_Name'] and r['Counter_Name']==C]
    # the 64-caller phase: launches with more than one list (Grid_Size_Y > 1) ... keep all, and the multi-list ones apart
    tot=sum(float(r['Counter_Value']) for r in rows)
    multi=[r for r in rows if int(r.get('Grid_Size_Y', r.get('Grid_Size','1')) or 1)>1]
    out[C]={"dispatches":len(rows),"kb_total":tot,"cols":list(rows[0].keys())[:24]}
    log=open('gpurun_out/pmc_ranked10_%s.log'%C).read()
    out[C]["log_tail"]=log[-600:]
json.dump(out,open('gpurun_out/r3_pmc_ranked10_summary.json','w'),indent=1)
print(json.dumps(out)[:3000])
PY
