#!/bin/bash
mkdir -p gpurun_out
timeout 300 tools/bin/ranked_bench 2000000 200000 3 32 16 64 > gpurun_out/p11_ranked_2m.jsonl 2> gpurun_out/p11_ranked_2m.err; echo rc=$?
timeout 400 tools/bin/ranked_bench 10000000 200000 3 24 1 16 64 > gpurun_out/p11_ranked_10m.jsonl 2> gpurun_out/p11_ranked_10m.err; echo rc=$?
sed 's/"config.*"queries_per_s"/"qps"/' gpurun_out/p11_ranked_2m.jsonl gpurun_out/p11_ranked_10m.jsonl | cut -c1-100
for i in 1 2 3; do
MSI_VM_PROFILE=1 timeout 400 tools/bin/ranked_bench 10000000 200000 3 16 16 > gpurun_out/p11_prof10_$i.jsonl 2> gpurun_out/p11_prof10_$i.err; echo "prof $i rc=$?"; grep "msi_vm profile" gpurun_out/p11_prof10_$i.err; grep -c "fewer documents" gpurun_out/p11_prof10_$i.err
done
