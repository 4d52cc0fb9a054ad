# Round 5: the fused-launch collapse of round 4, reproduced and bounded.  Budget of waiting workgroups in flight as
# a share of the device's residency (MSI_VM_FUSED_WGS_PCT: 400 = round 4's 4 096 on 1 024 resident workgroups; 100; 25 = the
# default since round 5), lists of up to 24 / 160 chunks fused; MSI_VM_PROFILE prints how long the waiters waited.
set -x
mkdir -p gpurun_out
for PCT in 400 100 25; do
  echo "== MSI_VM_FUSED_WGS_PCT=$PCT"
  MSI_VM_FUSED_WGS_PCT=$PCT MSI_VM_PROFILE=1 BURST=24,160 timeout 400 python tools/probes/r4_kw_classes.py 2>&1 | grep -v "amdgpu.ids\|first pass" | tail -6
done
