tag=${1:-r06w}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R
export LIO_SYNTH_CACHE=/tmp/lio_synth; mkdir -p $LIO_SYNTH_CACHE
export LIO_BENCH_NO_CALLBACK=1 LIO_BENCH_LONG_STEPS=0
for i in 1 2; do for v in 0 2 4 8; do
LIO_SIDE_CU_STRIDE=$v timeout 150 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --long-steps 0 --other-configs "" --no-loop-body > $O/b.json 2> $O/b.err
python - <<PY
import json
try:
    d=json.loads(open('$O/b.json').read().strip().splitlines()[-1])
    ew=d['node_loop_emulated_world8']; nl=d['node_loop']
    print('side stream on every ${v}th CU run $i | node_loop %.0f | world-8 deferred %.3f (step %.1f + update %.1f us = %.0f per rank)' % (nl['value'],
          ew['vs_node_loop_world1'], ew['scan_step_ms']*1e3, ew['map_update_ms_per_step']*1e3, ew['value']))
except Exception as e:
    print('stride $v failed', e, open('$O/b.err').read()[-300:])
PY
done; done
