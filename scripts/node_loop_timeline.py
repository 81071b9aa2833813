"""Kernel time line of ONE step of bench.py's node loop (scan step + lio_map_incremental) out of a rocprofv3 kernel trace: from the
first k-NN launch of a step to the launch before the next step's first k-NN, every launch with its start offset, the gap to the
previous one and its duration.
    python scripts/node_loop_timeline.py <dir with *kernel_trace.csv> <out.txt> [which step, counted from the end: default 6]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'] for r in rows]
lists = [i for i, n in enumerate(names) if 'map_incremental_lists_kernel' in n]
k = lists[-int(sys.argv[3]) if len(sys.argv) > 3 else -6]
a = k
while a > 0 and not ('knn27_kernel' in names[a] and ', false, ' in names[a].split('knn27_kernel')[1][:40] and 'cell_apply' in ' '.join(names[max(0, a - 6):a])):
    a -= 1
e = k
while e + 1 < len(rows) and 'knn27_kernel' not in names[e + 1]:
    e += 1
t0 = int(rows[a]['Start_Timestamp'])
prev = t0
out = open(sys.argv[2], 'w')
for r in rows[a:e + 1]:
    s, en = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    out.write('%9.1f us  +gap %6.1f  dur %6.1f  %s\n' % ((s - t0) / 1e3, (s - prev) / 1e3, (en - s) / 1e3, r['Kernel_Name'][:110]))
    prev = en
out.write('launches of the map update (from map_incremental_lists_kernel on): %d\n' % (e - k + 1))
out.close()
print(open(sys.argv[2]).read()[-6000:])
