# Round 5, one GPU-box visit: the -m gpu suite, the default bench line (driver's command), C5 with its reference-tree parity.
#   gpurun --timeout 1500 -- 'bash scripts/r05_check.sh r05a [c5]'
tag=${1:-r05a}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R
export LIO_SYNTH_CACHE=/tmp/lio_synth; mkdir -p $LIO_SYNTH_CACHE
T0=$(date +%s); stamp() { echo "== $1: $(( $(date +%s) - T0 )) s"; }
(timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | grep -a -v "thread\|amdgpu" | tail -25) > $O/pytest_gpu.log; tail -6 $O/pytest_gpu.log
stamp pytest
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_c2_driver.json 2> $O/bench_c2_driver.err; tail -1 $O/bench_c2_driver.json | cut -c1-300; tail -3 $O/bench_c2_driver.err
python - <<PY
import json
try:
    d=json.loads(open('$O/bench_c2_driver.json').read().strip().splitlines()[-1])
    print('value', round(d['value'],1), 'long', d.get('value_long_window'))
    print('other', json.dumps(d.get('other_configs'))[:1500])
    print('parity', json.dumps(d.get('parity',{}).get('neighbors_bit_exact'))[:800])
    print('cb', {k: round(v['value'],1) for k,v in d.items() if isinstance(v,dict) and 'value' in v and k.startswith('value_callback')})
except Exception as e: print('bench parse failed', e)
PY
stamp "bench c2 driver cmdline"
if [ "$2" = "c5" ]; then
timeout 1300 python bench.py --config c5 --steps 40 --warmup 5 > $O/bench_c5.json 2> $O/bench_c5.err; tail -1 $O/bench_c5.json | cut -c1-200; tail -3 $O/bench_c5.err
python - <<PY
import json
try:
    d=json.loads(open('$O/bench_c5.json').read().strip().splitlines()[-1])
    print('c5 value', round(d['value'],1), 'parity', json.dumps(d.get('parity'))[:1500])
except Exception as e: print('bench c5 parse failed', e)
PY
stamp "bench c5"
fi
