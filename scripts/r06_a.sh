# Round 6, first GPU visit: -m gpu suite, the 128-lane measurement kernel A/B'd against the 256-lane one on ONE box
# (LIO_RESIDUAL_BS), the re-ordered default bench line with `summary` at its end and `value_varied_priors`.
#   gpurun --timeout 1500 -- 'bash scripts/r06_a.sh r06a'
tag=${1:-r06a}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R
export LIO_SYNTH_CACHE=/tmp/lio_synth; mkdir -p $LIO_SYNTH_CACHE
T0=$(date +%s); stamp() { echo "== $1: $(( $(date +%s) - T0 )) s"; }
(timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | grep -a -v "thread\|amdgpu" | tail -15) > $O/pytest_gpu.log; tail -4 $O/pytest_gpu.log
stamp pytest
(LIO_RESIDUAL_BS=128 timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -q -x -k "not c5" 2>&1 | grep -a -v "thread\|amdgpu" | tail -8) > $O/pytest_bs128.log; tail -3 $O/pytest_bs128.log
stamp "pytest bs128"
python - <<PY > $O/residual_bs_ab.log 2>&1
import json, subprocess, sys, os
for cfg in ("c2", "c3"):
    for rep in range(3):
        for name, bs in (("bs256", ""), ("bs128", "128")):
            env = dict(os.environ)
            if bs: env["LIO_RESIDUAL_BS"] = bs
            out = subprocess.run([sys.executable, "bench.py", "--config", cfg, "--steps", "1000", "--warmup", "50", "--no-cpu-baseline", "--no-extras"],
                                 env=env, capture_output=True, text=True).stdout.strip().splitlines()[-1]
            d = json.loads(out)
            print(cfg, "rep", rep, name, "value %.0f  step %.2f us  p50 %.2f  kernels %s" % (d["value"], d["ms_per_step"] * 1e3, d["latency"]["p50"] * 1e3,
                  {k: round(v, 2) for k, v in d["kernels_us_all_launches_stamped"].items()}), flush=True)
PY
cat $O/residual_bs_ab.log
stamp "residual bs A/B"
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_c2_driver.json 2> $O/bench_c2_driver.err; tail -c 3000 $O/bench_c2_driver.json; tail -3 $O/bench_c2_driver.err
stamp "bench c2 driver cmdline"
