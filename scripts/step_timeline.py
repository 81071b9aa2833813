"""Kernel time line of the last few scan steps of bench.py's CONTRACT LOOP (--no-extras) out of a rocprofv3 kernel trace: start
offset, gap to the previous kernel's end, duration.  A scan step at C2 = first search + model, model, bounded search + model, model;
the copy of the next scan rides on the copy queue.
    python scripts/step_timeline.py <dir with *kernel_trace.csv> <out.txt> [steps to print: default 3] [steps to skip at the end: default 60]
(the sampled repeat behind the timed region stamps every launch with events: skip it)"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
nsteps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
skip = int(sys.argv[4]) if len(sys.argv) > 4 else 60
firsts = [i for i, r in enumerate(rows) if 'knn27_kernel<8, false, false' in r['Kernel_Name']]
b = firsts[-skip]
a = firsts[-skip - nsteps]
t0 = int(rows[a]['Start_Timestamp'])
prev_end = None
out = open(sys.argv[2], 'w')
gaps = 0.0
for r in rows[a:b]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
    gaps += max(gap, 0.0)
    prev_end = max(e, prev_end or e)
    out.write('%9.1f us  +gap %6.1f  dur %6.1f  %s\n' % ((s - t0) / 1e3, gap, (e - s) / 1e3, r['Kernel_Name'][:100]))
span = (int(rows[b]['Start_Timestamp']) - t0) / 1e3
out.write('%d steps: %.1f us per step, %.1f us of it between kernels\n' % (nsteps, span / nsteps, gaps / nsteps))
out.close()
print(open(sys.argv[2]).read())
