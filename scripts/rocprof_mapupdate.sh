# rocprofv3 kernel-trace of bench.py --map-update; stats in gpurun_out/$1/
tag=${1:-r01h}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/$tag
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$tag/trace -- python $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline --map-update > $R/gpurun_out/$tag/bench.log 2> $R/gpurun_out/$tag/bench.err
find $R/gpurun_out/$tag/trace -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/$tag/kernel_stats.csv \;
find $R/gpurun_out/$tag/trace -name "*kernel_trace.csv" -delete
python - <<PY
import csv,re
rows=list(csv.DictReader(open("$R/gpurun_out/$tag/kernel_stats.csv")))
def short(n):
    n=re.sub(r"rocprim::ROCPRIM_\d+_NS::","rocprim::",n)
    if n.startswith("void rocprim"):
        m=re.findall(r"rocprim::detail::(radix_sort_onesweep\w*|\w+_kernel\w*|lookback_scan\w*|partition\w*|merge\w*|scan\w*|reduce_by_key\w*|\w+_impl\w*)",n)
        return "rocprim::"+(m[1] if len(m)>1 else (m[0] if m else "kernel"))
    return n[:90]
for r in rows[:40]:
    print("%-92s %6s %10.1f %9.2f"%(short(r["Name"]),r["Calls"],float(r["TotalDurationNs"])/1e3,float(r["AverageNs"])/1e3))
PY
tail -1 $R/gpurun_out/$tag/bench.log | cut -c1-200
