"""Kernel time line of ONE timed segment of bench.py's emulated world (the last one of the run) out of a rocprofv3 kernel trace:
from the end of the last virtual rank's parking (replicate_stage_kernel) to the end of the timed rank's exchange.  With the
deferred schedule the plan of the held batch runs on a second queue beside the scan step: the queue of every kernel is printed.
    python scripts/emulated_timeline.py <dir with *kernel_trace.csv> <out.txt> [which exchange, counted from the end: default 5]
(bench.py --steps 8 runs 12 exchanges under the immediate schedule, then 12 under the deferred one; the last two of each are the
forced-rebuild steps of `sustained`)"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'] for r in rows]
qcol = 'Queue_Id' if 'Queue_Id' in rows[0] else None
compacts = [i for i, n in enumerate(names) if 'replicate_compact_kernel' in n]
last_compact = compacts[-int(sys.argv[3]) if len(sys.argv) > 3 else -5]
a = last_compact
while a > 0 and 'replicate_stage_kernel' not in names[a]:
    a -= 1
a += 1
e = last_compact
while e + 1 < len(rows) and 'knn27' not in names[e + 1] and 'replicate_stage_kernel' not in names[e + 1]:
    e += 1
t0 = int(rows[a]['Start_Timestamp'])
qs = {}
out = open(sys.argv[2], 'w')
ends = {}
for r in rows[a:e + 1]:
    s, en = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    q = qs.setdefault(r[qcol], len(qs)) if qcol else 0
    gap = (s - ends[q]) / 1e3 if q in ends else 0.0
    ends[q] = en
    out.write('%9.1f us  queue %d  +gap %6.1f  dur %6.1f  %s\n' % ((s - t0) / 1e3, q, gap, (en - s) / 1e3, r['Kernel_Name'][:110]))
out.close()
print(open(sys.argv[2]).read()[-7000:])
