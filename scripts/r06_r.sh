tag=${1:-r06r}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R
export LIO_SYNTH_CACHE=/tmp/lio_synth; mkdir -p $LIO_SYNTH_CACHE
export LIO_BENCH_NO_CALLBACK=1 LIO_BENCH_LONG_STEPS=0
for i in 1 2; do for v in "1 1" "0 1" "1 0" "0 0"; do
set -- $v
LIO_SIDE_EVENT=$1 LIO_PUBLISH_IN_KERNEL=$2 timeout 150 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --long-steps 0 --other-configs "" --no-loop-body > $O/b.json 2> $O/b.err
python - <<PY
import json
d=json.loads(open('$O/b.json').read().strip().splitlines()[-1])
ew=d['node_loop_emulated_world8']; nl=d['node_loop']
print('side event $1 publish in kernel $2 run $i | node_loop %.0f | world-8 deferred %.3f (step %.1f + update %.1f us) immediate %.3f (update %.1f us)' % (nl['value'],
      ew['vs_node_loop_world1'], ew['scan_step_ms']*1e3, ew['map_update_ms_per_step']*1e3, ew['immediate_schedule']['vs_node_loop_world1'], ew['immediate_schedule']['map_update_ms_per_step']*1e3))
PY
done; done
python - <<PY
import json, subprocess, sys, os
for cfg in ("c2",):
    for rep in range(3):
        for name, lib in (("round5", "$R/better_fastlio2_amd/lib/ab/liblio_hip_base.so"), ("round6", "")):
            env = dict(os.environ)
            if lib: env["LIO_HIP_LIBRARY"] = lib
            out = subprocess.run([sys.executable, "bench.py", "--config", cfg, "--steps", "2000", "--warmup", "50", "--no-cpu-baseline", "--no-extras"],
                                 env=env, capture_output=True, text=True).stdout.strip().splitlines()[-1]
            d = json.loads(out)
            print(cfg, "rep", rep, name, "value %.0f  step %.2f us  p50 %.2f  kernels %s" % (d["value"], d["ms_per_step"] * 1e3, d["latency"]["p50"] * 1e3,
                  {k: round(v, 2) for k, v in d["kernels_us_all_launches_stamped"].items()}), flush=True)
PY
