# A/B of two builds on ONE box through the contract loop: `value` and the per-family kernel times, alternating libraries.
#   gpurun --timeout 900 -- 'bash scripts/r05_value_ab.sh r05g'        (base: better_fastlio2_amd/lib/ab/liblio_hip_base.so)
tag=${1:-r05g}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R
export LIO_SYNTH_CACHE=/tmp/lio_synth; mkdir -p $LIO_SYNTH_CACHE
BASE=$R/better_fastlio2_amd/lib/ab/liblio_hip_base.so
python - <<PY > $O/value_ab.log 2>&1
import json, subprocess, sys, os
for cfg in ("c2", "c3"):
    for rep in range(3):
        for name, lib in (("base", "$BASE"), ("new ", "")):
            env = dict(os.environ)
            if lib: env["LIO_HIP_LIBRARY"] = lib
            out = subprocess.run([sys.executable, "bench.py", "--config", cfg, "--steps", "1000", "--warmup", "50", "--no-cpu-baseline", "--no-extras"],
                                 env=env, capture_output=True, text=True).stdout.strip().splitlines()[-1]
            d = json.loads(out)
            print(cfg, "rep", rep, name, "value %.0f  step %.2f us  p50 %.2f  kernels %s" % (d["value"], d["ms_per_step"] * 1e3, d["latency"]["p50"] * 1e3,
                  {k: round(v, 2) for k, v in d["kernels_us_all_launches_stamped"].items()}), flush=True)
PY
cat $O/value_ab.log
