tag=${1:-r06m}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R
export LIO_SYNTH_CACHE=/tmp/lio_synth; mkdir -p $LIO_SYNTH_CACHE
for i in 1 2; do for v in 0 1; do
LIO_PUBLISH_IN_KERNEL=$v LIO_BENCH_LONG_STEPS=0 timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --long-steps 0 --other-configs "" --no-loop-body > $O/bench_${v}_$i.json 2> $O/bench_${v}_$i.err
python - <<PY
import json
d=json.loads(open('$O/bench_${v}_$i.json').read().strip().splitlines()[-1])
ew=d['node_loop_emulated_world8']; nl=d['node_loop']
print('publish in kernel $v run $i value %.0f node_loop %.0f map_update %.1f us | ew deferred vs1 %.3f update %.1f us step %.1f us | immediate vs1 %.3f update %.1f us' % (d['value'], nl['value'], nl['map_update_ms_per_step']*1e3,
      ew['vs_node_loop_world1'], ew['map_update_ms_per_step']*1e3, ew['scan_step_ms']*1e3, ew['immediate_schedule']['vs_node_loop_world1'], ew['immediate_schedule']['map_update_ms_per_step']*1e3))
PY
done; done
( cd /tmp && export TMPDIR=/tmp && LIO_PUBLISH_IN_KERNEL=0 LIO_BENCH_LONG_STEPS=0 timeout 500 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --long-steps 0 --other-configs "" --no-loop-body > $O/trace_bench.log 2> $O/trace_bench.err )
python scripts/emulated_timeline.py $O/trace $O/emulated_timeline_immediate_publish_kernel.txt 17 | tail -8
find $O/trace -name "*kernel_trace.csv" -delete
