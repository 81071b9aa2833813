# One GPU-box visit: the -m gpu test-suite, the default bench, and a rocprofv3 kernel trace of the bench.
#   gpurun --timeout 900 -- 'bash scripts/gpu_check.sh r02b [pytest-args]'
tag=${1:-r02x}
shift
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/$tag
cd $R
(time python -m pytest tests -m gpu -x -q "$@") > gpurun_out/$tag/pytest_gpu.log 2>&1
grep -E "passed|failed|error" gpurun_out/$tag/pytest_gpu.log | tail -3
grep -E "^(FAILED|ERROR)|Error|assert" gpurun_out/$tag/pytest_gpu.log | head -20
python bench.py --steps 200 --warmup 20 > gpurun_out/$tag/bench.json 2> gpurun_out/$tag/bench.err
python - <<PY
import json
d=json.loads(open('gpurun_out/$tag/bench.json').read().strip().splitlines()[-1])
print('value', round(d['value'],1), 'ms/step', round(d['ms_per_step'],4), {k: round(v['value'],1) for k,v in d.items() if isinstance(v,dict) and 'value' in v and k!='cpu_baseline'})
print('kernels_us', {k: round(v,2) for k,v in d['kernels_us'].items()})
print('roofline', [(x['kernel'][:40], round(x['avg_launch_us'],2), round(x['frac'],3)) for x in d['roofline_detail']])
print('parity', d.get('parity',{}).get('neighbors_bit_exact'))
PY
bash scripts/rocprof_bench.sh $tag/prof > gpurun_out/$tag/rocprof.log 2>&1
head -14 gpurun_out/$tag/prof/kernel_stats.csv | cut -c1-150
