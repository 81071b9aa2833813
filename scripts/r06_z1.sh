tag=${1:-r06z1}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R
export LIO_SYNTH_CACHE=/tmp/lio_synth; mkdir -p $LIO_SYNTH_CACHE
export LIO_BENCH_NO_CALLBACK=1 LIO_BENCH_LONG_STEPS=0
for i in 1 2 3; do for v in occ4 occ5; do
if [ $v = occ4 ]; then export LIO_HIP_LIBRARY=$R/better_fastlio2_amd/lib/ab/liblio_hip_occ4.so; else unset LIO_HIP_LIBRARY; fi
timeout 150 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --long-steps 0 --other-configs "" --no-loop-body > $O/b.json 2> $O/b.err
python - <<PY
import json
d=json.loads(open('$O/b.json').read().strip().splitlines()[-1])
ew=d['node_loop_emulated_world8']; nl=d['node_loop']
print('fold_plan $v run $i | node_loop %.0f map update %.1f us | world-8 deferred %.3f (step %.1f + update %.1f us = %.0f per rank) immediate %.3f (update %.1f)' % (nl['value'], nl['map_update_ms_per_step']*1e3,
      ew['vs_node_loop_world1'], ew['scan_step_ms']*1e3, ew['map_update_ms_per_step']*1e3, ew['value'], ew['immediate_schedule']['vs_node_loop_world1'], ew['immediate_schedule']['map_update_ms_per_step']*1e3))
PY
done; done
