# The two C2 bench lines once more (bench.py gained the loop_body / pass_control keys behind the r05z set; the library is unchanged).
tag=${1:-r05z2}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R
export LIO_SYNTH_CACHE=/tmp/lio_synth; mkdir -p $LIO_SYNTH_CACHE
timeout 500 python bench.py > $O/bench_c2.json 2> $O/bench_c2.err; tail -1 $O/bench_c2.json | cut -c1-200
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_c2_driver.json 2> $O/bench_c2_driver.err; tail -1 $O/bench_c2_driver.json | cut -c1-200
python - <<PY
import json
for f in ('bench_c2.json','bench_c2_driver.json'):
    d=json.loads(open('$O/'+f).read().strip().splitlines()[-1])
    lb=d.get('loop_body',{})
    print(f, 'value', round(d['value']), 'long', round(d['value_long_window']['value']), 'frac', round(d['roofline']['frac'],3), 'emu', round(d['node_loop_emulated_world8']['vs_node_loop_world1'],3), 'loop', lb.get('staged_us',{}).get('total',{}).get('median'), lb.get('serial',{}).get('median_us'), lb.get('overlapped',{}).get('median_us'), d.get('pass_control'))
PY
