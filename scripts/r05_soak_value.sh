# Soak of the contract loop: many short bench runs, the slowest step and the pass-control counters of each (a stalled wait shows as a
# step of >= 10 ms and a non-zero gate_reenqueued).
tag=${1:-r05h}; cfgs=${2:-"c3 c2"}; runs=${3:-12}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R
export LIO_SYNTH_CACHE=/tmp/lio_synth; mkdir -p $LIO_SYNTH_CACHE
python - <<PY > $O/soak.log 2>&1
import json, subprocess, sys, os
for cfg in "$cfgs".split():
    for rep in range($runs):
        out = subprocess.run([sys.executable, "bench.py", "--config", cfg, "--steps", "1000", "--warmup", "50", "--no-cpu-baseline", "--no-extras"],
                             capture_output=True, text=True)
        try:
            d = json.loads(out.stdout.strip().splitlines()[-1])
            print(cfg, "rep", rep, "value %.0f  step %.2f us  p50 %.2f p99 %.2f max %.2f us" % (d["value"], d["ms_per_step"] * 1e3, d["latency"]["p50"] * 1e3, d["latency"]["p99"] * 1e3, d["latency"]["max"] * 1e3),
                  d["pass_control"], flush=True)
        except Exception as e:
            print(cfg, "rep", rep, "FAILED", e, out.stderr[-500:], flush=True)
PY
cat $O/soak.log
