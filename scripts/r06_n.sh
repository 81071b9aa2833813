tag=${1:-r06n}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R
export LIO_SYNTH_CACHE=/tmp/lio_synth; mkdir -p $LIO_SYNTH_CACHE
for v in oldkey new; do
if [ $v = oldkey ]; then export LIO_HIP_LIBRARY=$R/better_fastlio2_amd/lib/ab/liblio_hip_oldkey.so; else unset LIO_HIP_LIBRARY; fi
( cd /tmp && export TMPDIR=/tmp && LIO_BENCH_LONG_STEPS=0 timeout 500 rocprofv3 --kernel-trace --output-format csv -d $O/trace_$v -- python $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --long-steps 0 --other-configs "" --no-loop-body > $O/trace_$v.log 2> $O/trace_$v.err )
echo "== $v, immediate"; python scripts/emulated_timeline.py $O/trace_$v $O/emulated_timeline_immediate_$v.txt 17 | tail -15
echo "== $v, deferred"; python scripts/emulated_timeline.py $O/trace_$v $O/emulated_timeline_deferred_$v.txt | grep -a "queue 1\|lists_kernel"
find $O/trace_$v -name "*kernel_trace.csv" -delete
done
