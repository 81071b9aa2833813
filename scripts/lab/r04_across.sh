# round 4: the first search of the next scan armed across the scan boundary -- targeted tests, then A/B on the contract loop
# (off / armed behind a gate kernel / armed with the wait inside the search kernel; three alternations), then the step time line
tag=${1:-r04r}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R; export LIO_SYNTH_CACHE=/tmp/lio_synth
timeout 600 python -m pytest tests -m gpu -q -x -k "scan_boundary or enqueued_ahead or gate_gave_up or chain or smoke or scan_step" 2>&1 | grep -a -v "Rebuild thread\|Multi thread\|amdgpu.ids" | tail -15
for rep in 1 2 3; do for v in "0 3" "1 3" "1 7"; do
  set -- $v
  env LIO_ARM_NEXT_SCAN=$1 LIO_GATE_IN_KERNEL=$2 timeout 300 python bench.py --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline --no-extras > $O/ab_$1_$2_$rep.json 2> $O/ab_$1_$2_$rep.err || tail -3 $O/ab_$1_$2_$rep.err
  python - <<PY
import json
j=json.loads(open("$O/ab_$1_$2_$rep.json").read().strip().splitlines()[-1])
print("LIO_ARM_NEXT_SCAN=$1 LIO_GATE_IN_KERNEL=$2 rep $rep value %.0f step %.2f us p50 %.2f p99 %.2f kernels %s frac %.3f"%(j["value"],j["ms_per_step"]*1e3,j["latency"]["p50"]*1e3,j["latency"]["p99"]*1e3, {k:round(v,2) for k,v in j.get("kernels_us").items()}, j["roofline"]["frac"]))
PY
done; done | tee $O/ab.log
for v in 0 1; do env LIO_ARM_NEXT_SCAN=$v timeout 300 python bench.py --config c3 --steps 200 --warmup 20 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c3 LIO_ARM_NEXT_SCAN=$v value %.0f step %.2f us'%(j['value'],j['ms_per_step']*1e3))"; done | tee -a $O/ab.log
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extras > $O/trace_bench.json 2> $O/trace_bench.err )
python scripts/step_timeline.py $O/trace $O/step_timeline.txt 3 120
find $O/trace -name "*kernel_trace.csv" -delete
