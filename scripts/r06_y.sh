tag=${1:-r06y}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R
export LIO_SYNTH_CACHE=/tmp/lio_synth; mkdir -p $LIO_SYNTH_CACHE
T0=$(date +%s); stamp() { echo "== $1: $(( $(date +%s) - T0 )) s"; }
(timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | grep -a "passed\|failed\|Error\|assert \|^E " | tail -12) > $O/pytest_gpu.log; cat $O/pytest_gpu.log
stamp pytest
timeout 200 python tests/fuzz/fuzz_map.py 120 9500 > $O/fuzz_map.log 2>&1; echo "fuzz_map rc=$?"
timeout 300 python tests/fuzz/fuzz_loop.py 80 13500 > $O/fuzz_loop.log 2>&1; echo "fuzz_loop rc=$?"
stamp fuzz
export LIO_BENCH_NO_CALLBACK=1 LIO_BENCH_LONG_STEPS=0
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/bench.py --steps 16 --warmup 2 --no-cpu-baseline --long-steps 0 --other-configs "" --no-loop-body > $O/stats.log 2> $O/stats.err )
f=$(find $O/stats -name "*kernel_stats.csv" | head -1)
python - <<PY
import csv
rows = list(csv.DictReader(open('$f')))
for r in rows:
    n = r['Name']
    if any(k in n for k in ('cell_apply', 'cell_plan', 'fold_plan', 'fold_commit', 'pend_build', 'gsort', 'voxel_keys', 'rank_sort', 'map_incremental_lists', 'compact_dirty', 'replicate_compact')):
        print('%-46s calls %5s  avg %8.1f us  min %8.1f  max %8.1f' % (n[:46], r['Calls'], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3, float(r['MaxNs'])/1e3))
PY
find $O/stats -name "*kernel_trace.csv" -delete
stamp "kernel stats"
for i in 1 2; do
timeout 150 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --long-steps 0 --other-configs "" --no-loop-body > $O/b.json 2> $O/b.err
python - <<PY
import json
d=json.loads(open('$O/b.json').read().strip().splitlines()[-1])
ew=d['node_loop_emulated_world8']; nl=d['node_loop']
print('run $i | node_loop %.0f map update %.1f us | world-8 deferred %.3f (step %.1f + update %.1f us = %.0f per rank) immediate %.3f (update %.1f)' % (nl['value'], nl['map_update_ms_per_step']*1e3,
      ew['vs_node_loop_world1'], ew['scan_step_ms']*1e3, ew['map_update_ms_per_step']*1e3, ew['value'], ew['immediate_schedule']['vs_node_loop_world1'], ew['immediate_schedule']['map_update_ms_per_step']*1e3))
PY
done
