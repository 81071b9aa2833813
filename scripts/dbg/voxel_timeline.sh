# kernel time line of ONE voxel-filter call inside the loop body (rocprofv3 --kernel-trace):
#   gpurun --timeout 300 -- 'bash scripts/dbg/voxel_timeline.sh'
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/voxel_tl; rm -rf $O; mkdir -p $O
export LIO_SYNTH_CACHE=/tmp/lio_synth
timeout 200 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python $R/scripts/loop_body_step.py 8 > $O/run.log 2>&1
python - <<PY
import csv, glob
f = glob.glob('$O/trace/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'] for r in rows]
b = max(i for i, n in enumerate(names) if 'centroid_kernel' in n)
a = max(i for i in range(b) if 'undistort_kernel' in names[i])
t0 = int(rows[a]['Start_Timestamp']); prev = t0
for r in rows[a:b + 3]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print('%9.1f us  +gap %6.1f  dur %6.1f  %s' % ((s - t0) / 1e3, (s - prev) / 1e3, (e - s) / 1e3, r['Kernel_Name'][:90]))
    prev = e
PY
