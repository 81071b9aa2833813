# Round 6 against round 5 on ONE box: the round-5 build (ab/liblio_hip_base.so, made from commit 30a4eb9 before anything else this round)
# and the tree, alternating, through the same bench.py.
tag=${1:-r06p}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R
export LIO_SYNTH_CACHE=/tmp/lio_synth; mkdir -p $LIO_SYNTH_CACHE
for i in 1 2 3; do for v in round5 round6; do
if [ $v = round5 ]; then export LIO_HIP_LIBRARY=$R/better_fastlio2_amd/lib/ab/liblio_hip_base.so; else unset LIO_HIP_LIBRARY; fi
LIO_BENCH_NO_CALLBACK=1 LIO_BENCH_LONG_STEPS=0 timeout 150 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --long-steps 0 --other-configs "" > $O/bench_${v}_$i.json 2> $O/bench_${v}_$i.err
python - <<PY
import json
d=json.loads(open('$O/bench_${v}_$i.json').read().strip().splitlines()[-1])
ew=d['node_loop_emulated_world8']; nl=d['node_loop']; lb=d.get('loop_body', {})
print('$v run $i value %.0f | node_loop %.0f (p50 %.1f us, map update %.1f us) | world-8 deferred %.3f of N=1 (step %.1f + update %.1f us = %.0f scans/s per rank) immediate %.3f (update %.1f us, %.0f) | loop body serial %.1f overlapped %.1f us' % (d['value'], nl['value'], nl['latency']['p50']*1e3, nl['map_update_ms_per_step']*1e3,
      ew['vs_node_loop_world1'], ew['scan_step_ms']*1e3, ew['map_update_ms_per_step']*1e3, ew['value'], ew['immediate_schedule']['vs_node_loop_world1'], ew['immediate_schedule']['map_update_ms_per_step']*1e3, ew['immediate_schedule']['value'],
      lb.get('serial', {}).get('median_us', 0), lb.get('overlapped', {}).get('median_us', 0)))
PY
done; done
