# Kernel time line of ONE scan of scripts/loop_body_step.py (rocprofv3 --kernel-trace): every launch of the loop body in
# order, start offset and duration -- where the launches and the gaps between them are.
#   gpurun --timeout 600 -- 'bash scripts/map_timeline.sh r02x'
tag=${1:-r02x}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
timeout 400 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python $R/scripts/loop_body_step.py 12 > $O/loop.log 2> $O/loop.err
python - <<PY
import csv, glob
f = glob.glob('$O/trace/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'] for r in rows]
idx = [i for i, n in enumerate(names) if 'undistort_kernel' in n]
a, b = idx[len(idx) - 3], idx[len(idx) - 2]
t0 = int(rows[a]['Start_Timestamp'])
prev_end = t0
out = open('$O/timeline.txt', 'w')
for r in rows[a:b]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    line = '%9.1f us  +gap %6.1f  dur %6.1f  %s' % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3, r['Kernel_Name'][:90])
    print(line); out.write(line + '\n')
    prev_end = e
PY
rm -rf $O/trace
