cd $GRAFT_REPO_ROOT; export LIO_SYNTH_CACHE=/tmp/lio_synth; mkdir -p gpurun_out/r03z
T0=$(date +%s); stamp() { echo "== $1: $(( $(date +%s) - T0 )) s"; }
timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -q -x 2>&1 | grep -a -E "passed|failed|error|Error|assert" | tail -8
stamp "parity+fullsize tests"
timeout 200 python tests/fuzz/fuzz_measure.py 120 31000 2>&1 | grep -a -v thread | tail -2; echo "fuzz_measure rc=$?"
timeout 200 python tests/fuzz/fuzz_loop.py 60 32000 2>&1 | grep -a -v thread | tail -2
stamp "fuzz"
timeout 300 python bench.py --steps 100 --warmup 10 > gpurun_out/r03z/bench_c2.json 2> gpurun_out/r03z/bench_c2.err; tail -c 300 gpurun_out/r03z/bench_c2.err
stamp "bench c2"
timeout 300 python bench.py --config c3 --steps 100 --warmup 10 > gpurun_out/r03z/bench_c3.json 2> gpurun_out/r03z/bench_c3.err
stamp "bench c3"
python - <<'PY'
import json
for c in ("c2","c3"):
    d=json.loads(open('gpurun_out/r03z/bench_%s.json'%c).read().strip().splitlines()[-1])
    print(c,'value', round(d['value'],1), 'ms/step', round(d['ms_per_step'],4), {k: round(v['value'],1) for k,v in d.items() if isinstance(v,dict) and 'value' in v and k!='cpu_baseline'})
    print(' kernels_us', {k: round(v,2) for k,v in d['kernels_us'].items()})
    print(' roofline', [(x['kernel'][:44], round(x['avg_launch_us'],2), round(x['frac'],3), round(x.get('C_mean',0),1)) for x in d['roofline_detail']], d['roofline']['traffic_note'])
    for k in d:
        if k.startswith('node_loop'): print(' ',k, {kk: (round(vv,4) if isinstance(vv,float) else vv) for kk,vv in d[k].items() if kk not in ('latency','how')})
PY
