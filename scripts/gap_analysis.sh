# kernel-to-kernel gaps inside scan steps, from a rocprofv3 kernel trace of the default bench
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/gaps; mkdir -p $R/gpurun_out/gaps
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/gaps/trace -- python $R/bench.py --steps 60 --warmup 5 --no-cpu-baseline > /dev/null 2>&1
python - <<PY
import csv, glob, collections
f = glob.glob("$R/gpurun_out/gaps/trace/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    for k in ("knn27", "knn_tree", "residual_kernel<false, true", "residual_kernel<false, false", "transform"):
        if k in n: return k
    return None
ev = [(short(r["Kernel_Name"]), int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows]
ev = [e for e in ev if e[0]]
ev = ev[len(ev) // 2:]                      # the timed region (later half)
gaps = collections.defaultdict(list)
for a, b in zip(ev, ev[1:]):
    gaps[(a[0], b[0])].append((b[1] - a[2]) / 1e3)
dur = collections.defaultdict(list)
for e in ev: dur[e[0]].append((e[2] - e[1]) / 1e3)
for k, v in sorted(dur.items()): print("dur  %-34s n=%4d avg %.2f us" % (k, len(v), sum(v) / len(v)))
for k, v in sorted(gaps.items()):
    v2 = sorted(v); print("gap  %-34s -> %-34s n=%4d median %.2f us  avg %.2f" % (k[0], k[1], len(v), v2[len(v2) // 2], sum(v) / len(v)))
PY
rm -rf $R/gpurun_out/gaps/trace
