# Final evidence set of round 6, second half (rebuild by cell runs, fused scan preparation): scripts/end_of_round_r06.sh + the A/Bs of
# the two changes on the same box.   gpurun --timeout 3300 -- 'bash scripts/end_of_round_r06b.sh r07z'; then
# `python scripts/collect_profiles.py r07z` (+ the extra logs named in profiles/README.md).
tag=${1:-r07z}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R
bash scripts/end_of_round_r06.sh $tag
export LIO_SYNTH_CACHE=/tmp/lio_synth
for v in 1 0 1 0; do LIO_REBUILD_BY_CELLS=$v timeout 300 python scripts/rebuild_ab.py 2>&1 | tail -1 | tee -a $O/rebuild_ab.log; done
for v in 1 0 1 0; do echo "LIO_PREP_FUSED=$v" | tee -a $O/loopbody_prep_fused_ab.log; LIO_PREP_FUSED=$v timeout 300 python scripts/loop_body_step.py 60 2>&1 | grep -a -E "imu\+undistort|voxel  |SERIAL |OVERLAPPED" | cut -c1-170 | tee -a $O/loopbody_prep_fused_ab.log; done
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace_rebuild -- python $R/scripts/rebuild_ab.py 5000000 4 > $O/trace_rebuild.log 2>&1 )
python - <<PY
import csv, glob
f = glob.glob('$O/trace_rebuild/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'] for r in rows]
k = a = [i for i, n in enumerate(names) if 'cell_runs_count_kernel' in n][-1]
while a > 0 and 'live_tile_count' not in names[a]: a -= 1
e = k
while e + 1 < len(rows) and 'fill_child_blocks' not in names[e]: e += 1
e = min(e + 14, len(rows) - 1)
t0 = int(rows[a]['Start_Timestamp']); prev = t0
out = open('$O/rebuild_timeline.txt', 'w')
out.write('# one full rebuild of the C2 map (5.1 M live points, 1.68 M cells) by cell runs: rocprofv3 --kernel-trace of scripts/rebuild_ab.py, from the compaction to the last tree level\n')
for r in rows[a:e + 1]:
    s, en = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    out.write('%9.1f us  +gap %6.1f  dur %6.1f  %s\n' % ((s - t0) / 1e3, (s - prev) / 1e3, (en - s) / 1e3, r['Kernel_Name'][:120]))
    prev = en
out.close()
print(open('$O/rebuild_timeline.txt').read()[-3000:])
PY
find $O/trace_rebuild -name "*kernel_trace.csv" -delete
