# what the step of the emulated world of eight that meets the rebuild budget spends its time on: kernel time line around the forced rebuild
tag=${1:-r07e}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R
export LIO_SYNTH_CACHE=/tmp/lio_synth; mkdir -p $LIO_SYNTH_CACHE
( cd /tmp && export TMPDIR=/tmp && LIO_TRACE_REBUILDS=1 LIO_BENCH_LONG_STEPS=0 LIO_BENCH_NO_CALLBACK=1 timeout 500 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --long-steps 0 --other-configs "" --no-loop-body > $O/trace_bench.log 2> $O/trace_bench.err )
grep -a "lio:" $O/trace_bench.err | tail -8 | cut -c1-250
python - <<PY
import csv, glob
f = glob.glob('$O/trace/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'] for r in rows]
idx = [i for i, n in enumerate(names) if 'cell_runs_count_kernel' in n]
print('rebuilds by cell runs in the trace:', len(idx))
k = idx[-1]
t0 = int(rows[k]['Start_Timestamp'])
qs = {}
out = open('$O/world_rebuild_timeline.txt', 'w')
prev = None
for r in rows:
    s, en = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    if s < t0 - 400000 or s > t0 + 2200000: continue
    q = qs.setdefault(r.get('Queue_Id', '0'), len(qs))
    out.write('%9.1f us  queue %d  +gap %6.1f  dur %6.1f  %s\n' % ((s - t0) / 1e3, q, (s - prev) / 1e3 if prev else 0.0, (en - s) / 1e3, r['Kernel_Name'][:100]))
    prev = en
out.close()
print(open('$O/world_rebuild_timeline.txt').read()[:14000])
PY
find $O/trace -name "*kernel_trace.csv" -delete
