# the other BASELINE configs through bench.py (C3: HAP 24 k vs 2 M; C5: 1 M vs 50 M), JSON lines kept for profiles/
tag=${1:-r03x}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R
timeout 600 python bench.py --config c3 --steps 200 --warmup 20 > $O/bench_c3.json 2> $O/bench_c3.err
timeout 1500 python bench.py --config c5 --steps 20 --warmup 4 --no-cpu-baseline > $O/bench_c5.json 2> $O/bench_c5.err
python - <<PY
import json
for cfg in ('c3', 'c5'):
    try:
        d = json.loads(open('$O/bench_%s.json' % cfg).read().strip().splitlines()[-1])
    except Exception as e:
        print(cfg, 'no line', e); continue
    print(cfg, 'value', round(d['value'], 1), 'kernels_us', {k: round(v, 2) for k, v in d['kernels_us'].items()},
          [(x['kernel'][:28], round(x['avg_launch_us'], 2), round(x['frac'], 3)) for x in d['roofline_detail']],
          {k: (round(v['value'], 1), v.get('sustained', {}).get('value')) for k, v in d.items() if k.startswith('node_loop')},
          'parity', d['parity'].get('pos_diff_m'), d['parity'].get('neighbors_bit_exact'))
PY
