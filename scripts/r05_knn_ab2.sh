# A/B on ONE box (second k-NN question of round 5): the previous neighbours of a bounded search asked for at kernel ENTRY (beside the
# transform) against behind the barrier; then the default bench line (with its loop_body block) and the -m gpu suite.
tag=${1:-r05f}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R
export LIO_SYNTH_CACHE=/tmp/lio_synth; mkdir -p $LIO_SYNTH_CACHE
BASE=$R/better_fastlio2_amd/lib/ab/liblio_hip_base.so
for rep in 1 2 3; do
  echo "== base, repetition $rep"
  AB_LANES=8,16 AB_WALKS=4 AB_SIZES=100000,24000,8600 LIO_HIP_LIBRARY=$BASE timeout 300 python scripts/dbg/knn_walk_ab.py c2 2>&1 | grep -a "nq"
  echo "== new, repetition $rep"
  AB_LANES=8,16 AB_WALKS=4 AB_SIZES=100000,24000,8600 timeout 300 python scripts/dbg/knn_walk_ab.py c2 2>&1 | grep -a "nq"
done > $O/knn_ab.log 2>&1
cat $O/knn_ab.log
(AB_LANES=8 AB_WALKS=4 LIO_HIP_LIBRARY=$BASE timeout 200 python scripts/dbg/knn_walk_ab.py c3 2>&1 | grep -a "nq" | sed 's/^/base /'; AB_LANES=8 AB_WALKS=4 timeout 200 python scripts/dbg/knn_walk_ab.py c3 2>&1 | grep -a "nq" | sed 's/^/new  /') > $O/knn_ab_c3.log 2>&1
cat $O/knn_ab_c3.log
timeout 900 python -m pytest tests -m gpu -q 2>&1 | grep -a "passed\|failed\|error" | tail -3
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_c2_driver.json 2> $O/bench_c2_driver.err; tail -3 $O/bench_c2_driver.err
python - <<PY
import json
d=json.loads(open('$O/bench_c2_driver.json').read().strip().splitlines()[-1])
print('value', round(d['value']), 'long', round(d['value_long_window']['value']), 'loop_body', json.dumps(d.get('loop_body'))[:900])
print('kernels', d['kernels_us'], 'emu', d['node_loop_emulated_world8']['vs_node_loop_world1'])
PY
