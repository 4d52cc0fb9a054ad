# Round 6, eighth device call: what the command lists' workgroups spend their time on (MSI_VM_PROFILE) and how busy the device
# is during the keyword leg (kernel trace: vm_kernel's summed duration against the leg's wall time)
set -x
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
timeout 900 python -m pytest -x -q -m gpu tests/test_vs_gpu.py tests/test_zzz_vs_update_gpu.py tests/test_zz_group_gpu.py tests/test_rank_gpu.py 2>&1 | grep -a "passed\|failed\|error" | tail -3 | tee gpurun_out/r6_eighth_tests.log
MSI_VM_PROFILE=1 MSI_SEARCH_CPU_PROFILE=1 timeout 900 python tools/kw_leg.py --callers 256 --queries 3072 --fresh 4608 2>&1 | grep -a -v amdgpu.ids | tail -4 | cut -c1-2500 | tee gpurun_out/r6_vm_profile.log
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tr_kw
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr_kw -o tr -- python $R/tools/kw_leg.py --callers 256 --queries 3072 --fresh 4608 > /tmp/tr_kw.log 2>&1
F=$(find /tmp/tr_kw -name "*kernel_stats.csv" | head -1)
[ -n "$F" ] && cp $F $R/gpurun_out/r6_kw_leg_kernel_stats.csv && head -8 $F | cut -c1-200
grep -a "queries_per_s" /tmp/tr_kw.log | tail -1 | cut -c1-600 | tee $R/gpurun_out/r6_kw_leg_traced_line.log
# per-kernel timeline: how much of the measured window has a vm_kernel running, and how many at once
T=$(find /tmp/tr_kw -name "*kernel_trace.csv" | head -1)
python - "$T" <<'PY' | tee $R/gpurun_out/r6_kw_leg_device_busy.txt
import csv, sys
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        if "vm_kernel" in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Workgroup_Size"]) if "Workgroup_Size" in r else 0, int(r.get("Grid_Size", 0) or 0)))
rows.sort()
n = len(rows)
last = rows[int(n * 0.45):]          # the measured (fresh) pass is the last part of the process
t0, t1 = last[0][0], max(e for _, e, _, _ in last)
ev = []
for s, e, _, _ in last:
    ev.append((s, 1)); ev.append((e, -1))
ev.sort()
busy = 0; depth = 0; prev = t0; area = 0
for t, d in ev:
    if depth > 0: busy += t - prev
    area += depth * (t - prev)
    prev = t; depth += d
dur = [e - s for s, e, _, _ in last]
print(f"vm_kernel launches in the window: {len(last)}, window {1e-6*(t1-t0):.1f} ms, some vm_kernel running {100.0*busy/(t1-t0):.1f} % of it, "
      f"mean kernels in flight {area/(t1-t0):.2f}, mean kernel duration {1e-3*sum(dur)/len(dur):.1f} us, median {1e-3*sorted(dur)[len(dur)//2]:.1f} us, "
      f"mean grid size {sum(g for *_, g in last)/len(last):.0f} threads")
PY
