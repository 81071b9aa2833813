tag=${1:-r04z}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R; export LIO_SYNTH_CACHE=/tmp/lio_synth
( cd /tmp && export TMPDIR=/tmp && LIO_BENCH_LONG_STEPS=0 timeout 500 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline > $O/trace_bench.log 2> $O/trace_bench.err )
python scripts/emulated_timeline.py $O/trace $O/emulated_timeline_deferred.txt 5 | tail -45
python scripts/emulated_timeline.py $O/trace $O/emulated_timeline_immediate.txt 17 | tail -30
find $O/trace -name "*kernel_trace.csv" -delete
