# Round 6, GPU visit: the k-NN experiments.  (1) every group transforms its own query (small scans; LIO_KNN_TUNE=4 keeps the
# transforming wavefront) A/B'd through the contract loop at C3 and at a node's 8.6 k points; (2) what grouping the queries by cell
# is worth to the kernel as it is (scripts/knn_sorted_probe.py).
#   gpurun --timeout 1200 -- 'bash scripts/r06_f.sh r06f'
tag=${1:-r06f}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R
export LIO_SYNTH_CACHE=/tmp/lio_synth; mkdir -p $LIO_SYNTH_CACHE
T0=$(date +%s); stamp() { echo "== $1: $(( $(date +%s) - T0 )) s"; }
python - <<PY > $O/knn_own_transform_ab.log 2>&1
import json, subprocess, sys, os
for cfg in ("c3",):
    for rep in range(4):
        for name, v in (("transforming wavefront", "4"), ("own transform        ", "0")):
            env = dict(os.environ, LIO_KNN_TUNE=v)
            out = subprocess.run([sys.executable, "bench.py", "--config", cfg, "--steps", "1000", "--warmup", "50", "--no-cpu-baseline", "--no-extras"],
                                 env=env, capture_output=True, text=True).stdout.strip().splitlines()[-1]
            d = json.loads(out)
            print(cfg, "rep", rep, name, "value %.0f  step %.2f us  p50 %.2f  kernels %s" % (d["value"], d["ms_per_step"] * 1e3, d["latency"]["p50"] * 1e3,
                  {k: round(v, 2) for k, v in d["kernels_us_all_launches_stamped"].items()}), flush=True)
PY
cat $O/knn_own_transform_ab.log
stamp "own transform A/B"
timeout 300 python scripts/dbg/map_update_steps.py 2>&1 | grep -a "calls us\|   fast" > $O/map_update_steps.log; cat $O/map_update_steps.log
stamp "map update steps"
timeout 600 python scripts/knn_sorted_probe.py c2 c3 2>&1 | grep -a -v "thread\|amdgpu" > $O/knn_sorted_probe.log; cat $O/knn_sorted_probe.log
stamp "sorted probe"
(timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | grep -a -v "thread\|amdgpu" | tail -5) > $O/pytest_part.log; tail -3 $O/pytest_part.log
stamp "pytest (part)"
