tag=${1:-r06h}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R
export LIO_SYNTH_CACHE=/tmp/lio_synth; mkdir -p $LIO_SYNTH_CACHE
T0=$(date +%s); stamp() { echo "== $1: $(( $(date +%s) - T0 )) s"; }
for i in 1 2 3; do
LIO_BENCH_LONG_STEPS=0 timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --long-steps 0 --other-configs "" --no-loop-body > $O/bench_$i.json 2> $O/bench_$i.err
python - <<PY
import json
d=json.loads(open('$O/bench_$i.json').read().strip().splitlines()[-1])
ew=d['node_loop_emulated_world8']; nl=d['node_loop']
print('run $i value %.0f node_loop %.0f map_update %.1f us | ew deferred vs1 %.3f update %.1f us step %.1f us slowest %s | immediate vs1 %.3f update %.1f us slowest %s' % (d['value'], nl['value'], nl['map_update_ms_per_step']*1e3,
      ew['vs_node_loop_world1'], ew['map_update_ms_per_step']*1e3, ew['scan_step_ms']*1e3, ew['slowest_steps_ms'], ew['immediate_schedule']['vs_node_loop_world1'], ew['immediate_schedule']['map_update_ms_per_step']*1e3, ew['immediate_schedule']['slowest_steps_ms']))
PY
done
stamp "three default-ish lines"
