# rocprofv3 kernel-trace of the default bench command; CSV stats land in gpurun_out/$1/
tag=${1:-r01g}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/$tag
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$tag/trace -- python $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline > $R/gpurun_out/$tag/bench.log 2> $R/gpurun_out/$tag/bench.err
find $R/gpurun_out/$tag/trace -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/$tag/kernel_stats.csv \;
head -12 $R/gpurun_out/$tag/kernel_stats.csv
tail -1 $R/gpurun_out/$tag/bench.log | cut -c1-600
find $R/gpurun_out/$tag/trace -name "*kernel_trace.csv" -delete
