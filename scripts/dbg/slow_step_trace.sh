# kernel time line around the slowest fold_commit_kernel of a 200-step bench (the emulated-world window's outlier step)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/slow; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 500 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline > $O/b.log 2> $O/b.err
python - <<PY
import csv, glob
f = glob.glob('$O/trace/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
dur = lambda r: (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
cand = [i for i, r in enumerate(rows) if dur(r) > 300 and 'knn27_kernel<8, true' not in r['Kernel_Name'] and 'rocprim' not in r['Kernel_Name'] and 'copyBuffer' not in r['Kernel_Name']]
print('kernels longer than 300 us:', [(rows[i]['Kernel_Name'][:50], round(dur(rows[i]), 1)) for i in cand][:12])
for i in cand[:3]:
    a, b = max(0, i - 8), min(len(rows), i + 6)
    t0 = int(rows[a]['Start_Timestamp']); prev = t0
    for r in rows[a:b]:
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        print('%9.1f us  +gap %6.1f  dur %7.1f  %s' % ((s - t0) / 1e3, (s - prev) / 1e3, (e - s) / 1e3, r['Kernel_Name'][:90])); prev = e
    print()
PY
rm -rf $O/trace
