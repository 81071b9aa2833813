# round 4: the wait for the host's word inside the first kernel of an armed pass -- targeted tests, then A/B on the contract loop
# (LIO_GATE_IN_KERNEL = 0 gate kernel / 1 re-use passes / 3 all armed passes; three alternations)
tag=${1:-r04p}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R; export LIO_SYNTH_CACHE=/tmp/lio_synth
timeout 600 python -m pytest tests -m gpu -q -x -k "enqueued_ahead or gate_gave_up or deferred_schedule or chain or failed_pass or scan_set_async or smoke" 2>&1 | grep -a -v "Rebuild thread\|Multi thread\|amdgpu.ids" | tail -15
for rep in 1 2 3; do for v in 0 1 3; do
  env LIO_GATE_IN_KERNEL=$v timeout 300 python bench.py --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline --no-extras > $O/ab_${v}_$rep.json 2> $O/ab_${v}_$rep.err
  python - <<PY
import json
j=json.loads(open("$O/ab_${v}_$rep.json").read().strip().splitlines()[-1])
print("LIO_GATE_IN_KERNEL=$v rep $rep value %.0f step %.2f us p50 %.2f p99 %.2f kernels %s"%(j["value"],j["ms_per_step"]*1e3,j["latency"]["p50"]*1e3,j["latency"]["p99"]*1e3, j.get("kernels_us")))
PY
done; done | tee $O/ab.log
for v in 0 3; do env LIO_GATE_IN_KERNEL=$v timeout 300 python bench.py --config c3 --steps 200 --warmup 20 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c3 LIO_GATE_IN_KERNEL=$v value %.0f step %.2f us'%(j['value'],j['ms_per_step']*1e3))"; done | tee -a $O/ab.log
