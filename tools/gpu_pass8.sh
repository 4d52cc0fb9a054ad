#!/bin/bash
# one-off: where a workgroup's time goes inside vm_kernel (MSI_VM_PROFILE), 10 M and 2 M documents
mkdir -p gpurun_out
MSI_VM_PROFILE=1 timeout 400 tools/bin/ranked_bench 10000000 200000 3 16 1 16 > gpurun_out/p8_prof10.jsonl 2> gpurun_out/p8_prof10.err; echo rc=$?
grep "msi_vm profile" gpurun_out/p8_prof10.err
sed 's/"config.*"queries_per_s"/"qps"/' gpurun_out/p8_prof10.jsonl | cut -c1-120
MSI_VM_PROFILE=1 timeout 300 tools/bin/ranked_bench 2000000 200000 3 24 64 > gpurun_out/p8_prof2.jsonl 2> gpurun_out/p8_prof2.err; echo rc=$?
grep "msi_vm profile" gpurun_out/p8_prof2.err
sed 's/"config.*"queries_per_s"/"qps"/' gpurun_out/p8_prof2.jsonl | cut -c1-120
