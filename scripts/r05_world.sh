# A/B on ONE box: the commit of the deferred batch beside the next exchange (round 5) against the blocking settle (round 4), emulated
# world of 8 ranks, C2 and C3; then the replicate tests.
#   gpurun --timeout 1200 -- 'bash scripts/r05_world.sh r05e'
tag=${1:-r05e}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R
export LIO_SYNTH_CACHE=/tmp/lio_synth; mkdir -p $LIO_SYNTH_CACHE
timeout 600 python -m pytest tests/test_gpu_replicate.py tests/test_gpu_compact.py -q -x 2>&1 | tail -3
python - <<'PY' > $O/world_ab.log 2>&1
import json, subprocess, sys, os
for cfg in ("c2", "c3"):
    for rep in range(2):
        for mode in (0, 1):
            env = dict(os.environ, LIO_SETTLE_BESIDE=str(mode))
            out = subprocess.run([sys.executable, "bench.py", "--config", cfg, "--steps", "48", "--warmup", "8", "--no-cpu-baseline", "--long-steps", "0", "--other-configs", ""],
                                 env=env, capture_output=True, text=True).stdout.strip().splitlines()[-1]
            d = json.loads(out)
            e = d["node_loop_emulated_world8"]
            print(cfg, "rep", rep, "settle beside exchange" if mode else "blocking settle      ", "per rank %.0f scans/s = %.3f of N=1 (sustained %.3f); scan step %.3f ms, update %.3f ms, %d inserts" %
                  (e["value"], e["vs_node_loop_world1"], e["sustained"].get("vs_node_loop_world1", float("nan")), e["scan_step_ms"], e["map_update_ms_per_step"], e["inserted_per_step"]),
                  "| node_loop", round(d["node_loop"]["value"]), flush=True)
PY
cat $O/world_ab.log
