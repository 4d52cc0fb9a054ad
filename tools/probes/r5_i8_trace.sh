# kernel trace of the vector leg with the int8 level (768 queries per call, 10 M x 768)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/r5tr -o tr -- python $GRAFT_REPO_ROOT/tools/probes/r5_i8_variants.py > /tmp/r5tr.log 2>&1
F=$(find /tmp/r5tr -name "*kernel_stats.csv" | head -1)
cp $F $GRAFT_REPO_ROOT/gpurun_out/r5_i8_vector_leg_kernel_stats.csv
head -25 $F | cut -c1-220
grep -v amdgpu.ids /tmp/r5tr.log | tail -3
