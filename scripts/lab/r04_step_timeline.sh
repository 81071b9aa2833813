tag=${1:-r04q}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R; export LIO_SYNTH_CACHE=/tmp/lio_synth
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extras > $O/trace_bench.json 2> $O/trace_bench.err )
tail -c 300 $O/trace_bench.err
python scripts/step_timeline.py $O/trace $O/step_timeline.txt 3 120
find $O/trace -name "*kernel_trace.csv" -delete
