mkdir -p gpurun_out
MSI_VM_TRACE=gpurun_out/vm_trace.txt timeout 300 tools/bin/ranked_bench 2000000 200000 3 48 1 > gpurun_out/ranked_trace.jsonl 2>&1; echo rc=$?
wc -l gpurun_out/vm_trace.txt
sort -rn gpurun_out/vm_trace.txt | head -25
echo ...; sort -rn gpurun_out/vm_trace.txt | awk 'NR%400==0' | head -20
