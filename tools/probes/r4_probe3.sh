# round 4, third GPU pass: the typo matcher after the kernel split (occupancy 6 x 4 waves per CU)
set -x
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_dict_gpu.py tests/test_zz_fst_gpu.py "tests/test_configs_gpu.py" -m gpu -q -k "dict or fst or c3 or C3 or typo" 2>&1 | tail -4
timeout 120 python tools/fuzz_dict.py 700000 40 2>&1 | tail -2
for M in bits banded; do
  echo "== C3 $M"
  E=""; [ $M = banded ] && E="MSI_DICT_MATCHER=banded"
  env $E MSI_DICT_PROFILE=1 timeout 300 python bench.py --config c3 --steps 10 --warmup 2 --no-pmc --no-cpu-baseline > gpurun_out/r4_c3_${M}3.json 2> gpurun_out/r4_c3_${M}3.err; cut -c1-330 gpurun_out/r4_c3_${M}3.json; grep 'msi_dict profile' gpurun_out/r4_c3_${M}3.err
done
echo "== C3 bits, no profile, batch sizes"
for Q in 1536 8192 32768; do timeout 300 python bench.py --config c3 --steps 10 --warmup 2 --no-pmc --no-cpu-baseline --queries $Q 2>/dev/null | cut -c1-260; done
