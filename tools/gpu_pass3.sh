mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
timeout 600 python bench.py > gpurun_out/bench_c4.json 2> gpurun_out/bench_c4.err; echo "c4 rc=$?"; python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_c4.json'))
print({k:d[k] for k in ('value','ms_per_step')}, d['legs'], d['roofline']['frac'], d['parity']['mismatches'], d.get('cpu_baseline',{}).get('value'))
PY
tail -3 gpurun_out/bench_c4.err
timeout 300 python tools/bench_configs.py bq --rows 10000000 --dim 768 --reps 5 > gpurun_out/bq_10m.jsonl 2> gpurun_out/bq_10m.err; cat gpurun_out/bq_10m.jsonl; tail -2 gpurun_out/bq_10m.err
