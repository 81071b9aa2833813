# rocprofv3 kernel-trace of bench.py; CSV stats land in gpurun_out/$1/
#   kernel_stats.csv         the CONTRACT LOOP ONLY (--no-extras): the drive `value` is quoted on + the untimed event-stamped repeat of it.
#                            The averages of knn27_kernel<8,false,false,4> / <8,false,true,4> / residual_kernel here are the ones
#                            bench.py's roofline.avg_launch_us / kernels_us must agree with.
#   kernel_stats_all.csv     the default command with every extra (callback chain, node loop on a map that grows denser, emulated
#                            world, the 2 400-step window): per-kernel averages mix those regimes
tag=${1:-r01g}; cfg=${2:-c2}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/$tag
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$tag/trace -- python $R/bench.py --config $cfg --steps 100 --warmup 10 --no-cpu-baseline --no-extras > $R/gpurun_out/$tag/bench.log 2> $R/gpurun_out/$tag/bench.err
find $R/gpurun_out/$tag/trace -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/$tag/kernel_stats.csv \;
head -8 $R/gpurun_out/$tag/kernel_stats.csv
tail -1 $R/gpurun_out/$tag/bench.log | cut -c1-300
find $R/gpurun_out/$tag/trace -name "*kernel_trace.csv" -delete
if [ "${3:-all}" = "all" ]; then
rm -rf $R/gpurun_out/$tag/trace
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$tag/trace -- python $R/bench.py --config $cfg --steps 100 --warmup 10 --no-cpu-baseline > $R/gpurun_out/$tag/bench_all.log 2> $R/gpurun_out/$tag/bench_all.err
find $R/gpurun_out/$tag/trace -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/$tag/kernel_stats_all.csv \;
find $R/gpurun_out/$tag/trace -name "*kernel_trace.csv" -delete
fi
