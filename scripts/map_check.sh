# Map-maintenance check on the GPU box: gpu tests, the map / loop fuzzers (exit codes), loop-body timing, bench node_loop.
#   gpurun --timeout 1200 -- 'bash scripts/map_check.sh r02h'
tag=${1:-r02x}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -q 2>&1 | grep -a "passed\|failed\|error\|FAILED" | tail -8 > $O/pytest_gpu.log; cat $O/pytest_gpu.log
timeout 300 python tests/fuzz/fuzz_map.py > $O/fuzz_map.log 2>&1; echo "fuzz_map rc=$?"
timeout 300 python tests/fuzz/fuzz_loop.py > $O/fuzz_loop.log 2>&1; echo "fuzz_loop rc=$?"
timeout 200 python scripts/loop_body_step.py 2>&1 | grep -a -v "thread\|amdgpu" > $O/loopbody.log; tail -8 $O/loopbody.log
timeout 300 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err
python - <<PY
import json
d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1])
print('value', round(d['value'],1), {k: round(v['value'],1) for k,v in d.items() if isinstance(v,dict) and 'value' in v and k!='cpu_baseline'})
print({k: v for k, v in d['node_loop'].items() if k != 'value'})
PY
