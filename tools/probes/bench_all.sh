# every bench.py config on one GPU: the lines land under gpurun_out/ (copy the ones to keep into profiles/)
mkdir -p gpurun_out
for c in c4 c2 c3 c5; do
  timeout 900 python bench.py --config $c > gpurun_out/bench_$c.json 2> gpurun_out/bench_$c.err; echo "$c rc=$?"
  cut -c1-2500 gpurun_out/bench_$c.json; tail -3 gpurun_out/bench_$c.err
done
