# A/B of bench.py's contract loop on one box: plain candidate walk (LIO_KNN_WALK=0) vs the default launch shape, alternating.
#   gpurun --timeout 400 -- 'bash scripts/r03_value_ab.sh [config]'
cd $GRAFT_REPO_ROOT; export LIO_SYNTH_CACHE=/tmp/lio_synth
cfg=${1:-c2}
for rep in 1 2 3; do
  for w in 0 d; do
    if [ $w = d ]; then unset LIO_KNN_WALK; else export LIO_KNN_WALK=$w; fi
    timeout 200 python bench.py --config $cfg --steps 200 --warmup 20 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('walk $w  value %.0f  p50 %.1f us  knn first %.2f hinted %.2f  residual %.2f' % (d['value'], d['latency']['p50']*1e3, d['kernels_us']['knn27'], d['kernels_us'].get('knn27_hint',0), d['kernels_us']['residual']))"
  done
done
