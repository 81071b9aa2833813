tag=${1:-r06s}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R
export LIO_SYNTH_CACHE=/tmp/lio_synth; mkdir -p $LIO_SYNTH_CACHE
export LIO_BENCH_NO_CALLBACK=1 LIO_BENCH_LONG_STEPS=0
for v in round5 round6; do
if [ $v = round5 ]; then export LIO_HIP_LIBRARY=$R/better_fastlio2_amd/lib/ab/liblio_hip_base.so; else unset LIO_HIP_LIBRARY; fi
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$v -- python $R/bench.py --steps 16 --warmup 2 --no-cpu-baseline --long-steps 0 --other-configs "" --no-loop-body > $O/stats_$v.log 2> $O/stats_$v.err )
f=$(find $O/stats_$v -name "*kernel_stats.csv" | head -1)
echo "== $v"; python - <<PY
import csv
rows = list(csv.DictReader(open('$f')))
for r in rows:
    n = r['Name']
    if any(k in n for k in ('cell_apply', 'cell_plan', 'fold_plan', 'fold_commit', 'pend_build', 'gsort', 'voxel_keys', 'rank_sort', 'map_incremental_lists', 'compact_dirty', 'replicate_compact')):
        print('%-46s calls %5s  avg %8.1f us  min %8.1f  max %8.1f' % (n[:46], r['Calls'], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3, float(r['MaxNs'])/1e3))
PY
find $O/stats_$v -name "*kernel_trace.csv" -delete
done
