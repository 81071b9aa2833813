tag=${1:-r06x}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R
export LIO_SYNTH_CACHE=/tmp/lio_synth; mkdir -p $LIO_SYNTH_CACHE
export LIO_BENCH_NO_CALLBACK=1 LIO_BENCH_LONG_STEPS=0
for i in 1 2 3; do for v in 8192 16384 2048; do
LIO_RANK_MAX=$v timeout 150 python bench.py --gpus 1 --steps 48 --warmup 5 --no-cpu-baseline --long-steps 0 --other-configs "" --no-loop-body --emulate-world 0 > $O/b.json 2> $O/b.err
python - <<PY
import json
d=json.loads(open('$O/b.json').read().strip().splitlines()[-1])
nl=d['node_loop']
print('rank sort up to $v run $i | node_loop %.0f p50 %.1f us map update %.1f us inserted %.0f' % (nl['value'], nl['latency']['p50']*1e3, nl['map_update_ms_per_step']*1e3, nl['inserted_per_step']))
PY
done; done
timeout 300 python scripts/loop_body_step.py 40 2>&1 | grep -a "map_incremental\|SERIAL \|OVERLAPPED"
