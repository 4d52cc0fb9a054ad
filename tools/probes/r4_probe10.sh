set -x
timeout 120 python tools/fuzz_dict.py 810000 20 2>&1 | tail -1
for ST in 256 1024 100000; do
for Q in 1536 8192 32768; do echo "slice tiles $ST batch $Q"; MSI_DICT_SLICE_TILES=$ST timeout 300 python bench.py --config c3 --steps 10 --warmup 2 --no-pmc --no-cpu-baseline --queries $Q 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['cache_stream']['avg_launch_ms'])"; done; done
