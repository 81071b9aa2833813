# the emulated world of bench.py (deferred vs immediate exchange schedule) under an environment switch
tag=${1:-r04i}; var=${2:-LIO_SIDE_PRIORITY}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R; export LIO_SYNTH_CACHE=/tmp/lio_synth
for rep in 1 2; do for v in 0 1; do
  env $var=$v LIO_BENCH_LONG_STEPS=0 timeout 300 python bench.py --gpus 1 --steps 40 --warmup 10 --no-cpu-baseline > $O/w_${v}_$rep.json 2> $O/w_${v}_$rep.err
  python - <<PY
import json
j=json.loads(open("$O/w_${v}_$rep.json").read().strip().splitlines()[-1])
e=j["node_loop_emulated_world8"]; i=e["immediate_schedule"]
print("$var=$v rep $rep: node_loop %.0f | deferred %.0f (%.3f of N=1; scan %.1f us + update %.1f us; %d inserts) | immediate %.0f (%.3f; scan %.1f + update %.1f)"%(
  j["node_loop"]["value"], e["value"], e["vs_node_loop_world1"], e["scan_step_ms"]*1e3, e["map_update_ms_per_step"]*1e3, e["inserted_per_step"], i["value"], i["vs_node_loop_world1"], i["scan_step_ms"]*1e3, i["map_update_ms_per_step"]*1e3))
PY
done; done | tee $O/world.log
