set -x
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_dict_gpu.py tests/test_zz_fst_gpu.py tests/test_configs_gpu.py -m gpu -q -k "dict or fst or c3 or C3 or typo" 2>&1 | tail -3
timeout 120 python tools/fuzz_dict.py 800000 30 2>&1 | tail -1
for ST in 64 256 1024; do
for Q in 1536 8192 32768; do echo "slice tiles $ST batch $Q"; MSI_DICT_SLICE_TILES=$ST timeout 300 python bench.py --config c3 --steps 10 --warmup 2 --no-pmc --no-cpu-baseline --queries $Q 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['cache_stream']['avg_launch_ms'])"; done; done
MSI_DICT_PROFILE=1 timeout 300 python bench.py --config c3 --steps 10 --warmup 2 --no-pmc --no-cpu-baseline 2>&1 | grep 'msi_dict profile'
