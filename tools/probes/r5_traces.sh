# Round 5: kernel traces (rocprofv3 --kernel-trace --stats) of the C4 step and of every side configuration of the default command
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
for cfg in c4 c2 c3 c5; do
  extra=""
  [ $cfg = c4 ] && extra="--steps 6 --warmup 2 --no-also"
  rm -rf /tmp/tr_$cfg
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr_$cfg -o tr -- python $R/bench.py --config $cfg --no-pmc --no-cpu-baseline $extra > /tmp/tr_$cfg.log 2>&1
  F=$(find /tmp/tr_$cfg -name "*kernel_stats.csv" | head -1)
  [ -n "$F" ] && cp $F $R/gpurun_out/r5_bench_${cfg}_kernel_stats.csv && echo "== $cfg" && head -6 $F | cut -c1-200
  tail -1 /tmp/tr_$cfg.log | cut -c1-300
done
