# scan preparation with getMinMax3D inside the compensation kernel and the run heads inside the centroid kernel (LIO_PREP_FUSED=0: the
# launches of their own), the rebuild by cell runs again: parity tests, fuzz cases, A/Bs on one box
tag=${1:-r07c}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R
export LIO_SYNTH_CACHE=/tmp/lio_synth; mkdir -p $LIO_SYNTH_CACHE
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_compact.py -x -q -m gpu > $O/pytest.log 2>&1; grep -a -E "passed|failed|Error" $O/pytest.log | tail -5
timeout 300 python tests/fuzz/fuzz_map.py 150 17500 > $O/fuzz_map.log 2>&1; echo "fuzz_map rc=$?"; timeout 300 python tests/fuzz/fuzz_loop.py 100 21500 > $O/fuzz_loop.log 2>&1; echo "fuzz_loop rc=$?"
for v in 1 0 1 0; do LIO_REBUILD_BY_CELLS=$v timeout 300 python scripts/rebuild_ab.py 2>&1 | tail -1 | tee -a $O/rebuild_ab.log; done
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python $R/scripts/rebuild_ab.py 5000000 4 > $O/trace.log 2>&1 )
python - <<PY
import csv, glob
f = glob.glob('$O/trace/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'] for r in rows]
idx = [i for i, n in enumerate(names) if 'live_runs_count_kernel' in n]
if idx:
    k = a = idx[-1]
else:
    k = a = [i for i, n in enumerate(names) if 'cell_runs_count_kernel' in n][-1]
    while a > 0 and 'live_tile_count' not in names[a]: a -= 1
e = k
while e + 1 < len(rows) and 'fill_child_blocks' not in names[e]: e += 1
e = min(e + 14, len(rows) - 1)
t0 = int(rows[a]['Start_Timestamp']); prev = t0
out = open('$O/rebuild_timeline.txt', 'w')
for r in rows[a:e + 1]:
    s, en = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    out.write('%9.1f us  +gap %6.1f  dur %6.1f  %s\n' % ((s - t0) / 1e3, (s - prev) / 1e3, (en - s) / 1e3, r['Kernel_Name'][:120]))
    prev = en
out.close()
print(open('$O/rebuild_timeline.txt').read())
PY
find $O/trace -name "*kernel_trace.csv" -delete
