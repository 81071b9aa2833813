cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for L in 4 16; do
  mkdir -p $R/gpurun_out/lanes$L
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/lanes$L/trace -- python $R/bench.py --steps 40 --warmup 5 --no-cpu-baseline --lanes $L > $R/gpurun_out/lanes$L/bench.log 2> $R/gpurun_out/lanes$L/bench.err
  find $R/gpurun_out/lanes$L/trace -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/lanes$L/kernel_stats.csv \;
  echo "== lanes $L"; grep -a "knn27" $R/gpurun_out/lanes$L/kernel_stats.csv | cut -d, -f1-4 | cut -c1-120
  find $R/gpurun_out/lanes$L/trace -name "*kernel_trace.csv" -delete
done
