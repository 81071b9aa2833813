# bench time lines only (no tests): kernel trace of bench.py with the rebuild reasons on stderr
tag=${1:-r03x}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R
LIO_TRACE_REBUILDS=1 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
grep -a "full rebuild" $O/bench.err | tail -20
python - <<PY
import json
d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1])
print('value', round(d['value'],1), {k: round(v['value'],1) for k,v in d.items() if isinstance(v,dict) and 'value' in v and k!='cpu_baseline'})
for k in d:
    if k.startswith('node_loop'): print(k, d[k])
print('sustained', {k: d[k].get('sustained') for k in d if k.startswith('node_loop')})
PY
cd /tmp && export TMPDIR=/tmp
timeout 500 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline > $O/trace_bench.log 2> $O/trace_bench.err
python - <<PY
import csv, glob
f = glob.glob('$O/trace/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'] for r in rows]
idx = [i for i, n in enumerate(names) if 'replicate_compact_kernel' in n]
# one timed segment of the emulated world: from the timed rank's first k-NN launch before the last compact to the end of that apply
b = idx[-1]
a = b
while a > 0 and 'replicate_stage_kernel' not in names[a]: a -= 1
e = b
while e + 1 < len(rows) and 'knn27' not in names[e + 1]: e += 1
t0 = int(rows[a]['Start_Timestamp']); prev_end = t0
out = open('$O/emulated_timeline.txt', 'w')
for r in rows[a:e + 1]:
    s, en = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    line = '%9.1f us  +gap %6.1f  dur %6.1f  %s' % ((s - t0) / 1e3, (s - prev_end) / 1e3, (en - s) / 1e3, r['Kernel_Name'][:100])
    out.write(line + '\n'); prev_end = en
out.close()
print(open('$O/emulated_timeline.txt').read()[-6000:])
# one step of the N = 1 node loop: the last map update before the first replicate_stage_kernel
first_stage = next(i for i, n in enumerate(names) if 'replicate_stage_kernel' in n)
b = max(i for i in range(first_stage) if 'cell_apply_kernel' in names[i])
a = max(i for i in range(b) if 'map_incremental_lists_kernel' in names[i])
while a > 0 and not ('knn27_kernel<' in names[a] and ', false, false' in names[a]): a -= 1
t0 = int(rows[a]['Start_Timestamp']); prev_end = t0
out = open('$O/node_loop_timeline.txt', 'w')
for r in rows[a:b + 2]:
    s, en = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    line = '%9.1f us  +gap %6.1f  dur %6.1f  %s' % ((s - t0) / 1e3, (s - prev_end) / 1e3, (en - s) / 1e3, r['Kernel_Name'][:100])
    out.write(line + '\n'); prev_end = en
out.close()
print('---- N = 1 node loop step'); print(open('$O/node_loop_timeline.txt').read())
# one full rebuild inside the emulated loop (the last one): from its first compaction launch to the next k-NN
reb = [i for i, n in enumerate(names) if 'live_tile_count_kernel' in n or 'point_keys_kernel' in n]
if reb:
    a = reb[-1]
    while a > 0 and 'knn27' not in names[a - 1] and 'residual' not in names[a - 1] and 'cell_apply' not in names[a - 1] and 'pend_build' not in names[a - 1]: a -= 1
    e = reb[-1]
    while e + 1 < len(rows) and 'knn27' not in names[e + 1] and 'replicate_stage' not in names[e + 1]: e += 1
    t0 = int(rows[a]['Start_Timestamp']); prev_end = t0
    out = open('$O/rebuild_timeline.txt', 'w')
    for r in rows[a:e + 1]:
        s, en = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        out.write('%9.1f us  +gap %6.1f  dur %6.1f  %s\n' % ((s - t0) / 1e3, (s - prev_end) / 1e3, (en - s) / 1e3, r['Kernel_Name'][:110])); prev_end = en
    out.close()
    print('---- one full rebuild'); print(open('$O/rebuild_timeline.txt').read())
# per-kernel totals of the whole run
agg = {}
for r in rows:
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    k = r['Kernel_Name'][:120]
    c = agg.setdefault(k, [0, 0.0, 1e30, 0.0]); c[0] += 1; c[1] += d; c[2] = min(c[2], d); c[3] = max(c[3], d)
tot = sum(v[1] for v in agg.values())
with open('$O/kernel_stats.txt', 'w') as out:
    out.write('%-120s %8s %12s %10s %10s %10s %6s\n' % ('kernel', 'calls', 'total_us', 'avg_us', 'min_us', 'max_us', 'pct'))
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        out.write('%-120s %8d %12.1f %10.2f %10.2f %10.2f %6.2f\n' % (k, v[0], v[1], v[1] / v[0], v[2], v[3], 100 * v[1] / tot))
PY
rm -rf $O/trace
