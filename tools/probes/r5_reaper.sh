# Round 5, call 13: the reaper thread (MSI_VM_REAPER: completions noticed and searches woken by a second thread) against the
# single combiner of rounds 2-4, and 256 against 384 callers — one process, one index, one posting cache, every configuration
# on its own 2 304 fresh queries (tools/kw_leg.py --sweep); before it the command-list tests on the device with the reaper on
set -x
mkdir -p gpurun_out
(time timeout 80 python -m pytest tests/test_zz_vm_gpu.py tests/test_zz_levels_per_wait_gpu.py -m gpu -x -q) 2>&1 | tail -4
MSI_SEARCH_CPU_PROFILE=1 timeout 200 python tools/kw_leg.py --callers 384 --queries 3072 --segment 2304 \
  --sweep "1:256,0:256,1:384,0:384,1:256,0:256" 2>&1 | grep -v amdgpu.ids > gpurun_out/r5_reaper.jsonl
python - <<'P'
import json
for line in open("gpurun_out/r5_reaper.jsonl"):
    if not line.startswith("{"):
        print(line.rstrip()[:300]); continue
    d = json.loads(line)
    print("reaper", d["MSI_VM_REAPER"], "callers", d["callers"], "q/s", d["queries_per_s"], "cpus", d["host_cpus_used"], "p50", d["p50_ms_at_load"],
          "hit", d["posting_cache"]["hit_rate"], d["vm"], {k: d.get("host_cpu_us_per_query", {}).get(k) for k in ("search_threads", "list_submit_and_wait", "combiner_thread")})
P
