# round 4: A/B of an environment switch on the contract loop (three alternations), then the full check
tag=${1:-r04e}; var=${2:-LIO_GATE_MAIL_VRAM}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R; export LIO_SYNTH_CACHE=/tmp/lio_synth
for rep in 1 2 3; do for v in 0 1; do
  env $var=$v timeout 300 python bench.py --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline --no-extras > $O/ab_${v}_$rep.json 2> $O/ab_${v}_$rep.err
  python - <<PY
import json
j=json.loads(open("$O/ab_${v}_$rep.json").read().strip().splitlines()[-1])
print("$var=$v rep $rep value %.0f step %.2f us p50 %.2f p99 %.2f"%(j["value"],j["ms_per_step"]*1e3,j["latency"]["p50"]*1e3,j["latency"]["p99"]*1e3))
PY
done; done | tee $O/ab.log
