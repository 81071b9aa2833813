# Round 6, GPU visit: suite + fuzz seeds, node-loop A/B (round trip / fast), the default line (emulated world), time lines of the
# node loop and of one emulated-world exchange.
#   gpurun --timeout 1800 -- 'bash scripts/r06_d.sh r06d'
tag=${1:-r06d}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R
export LIO_SYNTH_CACHE=/tmp/lio_synth; mkdir -p $LIO_SYNTH_CACHE
T0=$(date +%s); stamp() { echo "== $1: $(( $(date +%s) - T0 )) s"; }
(timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | grep -a -v "thread\|amdgpu" | tail -25) > $O/pytest_gpu.log; tail -4 $O/pytest_gpu.log
stamp pytest
timeout 200 python tests/fuzz/fuzz_map.py 100 7500 > $O/fuzz_map.log 2>&1; echo "fuzz_map rc=$?"; tail -1 $O/fuzz_map.log
timeout 300 python tests/fuzz/fuzz_loop.py 60 11500 > $O/fuzz_loop.log 2>&1; echo "fuzz_loop rc=$?"; tail -1 $O/fuzz_loop.log
stamp fuzz
python - <<PY > $O/fast_update_ab.log 2>&1
import json, subprocess, sys, os
for cfg in ("c2", "c3"):
    for rep in range(2):
        for name, v in (("round trip", "0"), ("fast      ", "1")):
            env = dict(os.environ, LIO_FAST_UPDATE=v, LIO_BENCH_LONG_STEPS="0")
            out = subprocess.run([sys.executable, "bench.py", "--config", cfg, "--steps", "48", "--warmup", "5", "--no-cpu-baseline", "--long-steps", "0",
                                  "--other-configs", "", "--no-loop-body", "--emulate-world", "0"], env=env, capture_output=True, text=True).stdout.strip().splitlines()[-1]
            d = json.loads(out)
            nl = d["node_loop"]
            print(cfg, "rep", rep, name, "node_loop %.0f scans/s  step %.1f us  map update %.1f us  inserted %.0f  p50 %.1f  fast %s redone %s" %
                  (nl["value"], nl["ms_per_step"] * 1e3, nl["map_update_ms_per_step"] * 1e3, nl["inserted_per_step"], nl["latency"]["p50"] * 1e3,
                   nl.get("updates_without_a_host_round_trip"), nl.get("updates_redone_with_the_round_trip_after_a_list_outgrew_its_launches")), flush=True)
PY
cat $O/fast_update_ab.log
stamp "fast update A/B"
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_c2_driver.json 2> $O/bench_c2_driver.err; tail -c 2600 $O/bench_c2_driver.json; tail -3 $O/bench_c2_driver.err
stamp "bench c2 driver cmdline (no cpu baseline)"
( cd /tmp && export TMPDIR=/tmp && LIO_BENCH_LONG_STEPS=0 timeout 500 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --long-steps 0 --other-configs "" --no-loop-body > $O/trace_bench.log 2> $O/trace_bench.err )
python scripts/node_loop_timeline.py $O/trace $O/node_loop_timeline.txt 40 | tail -22
python scripts/emulated_timeline.py $O/trace $O/emulated_timeline.txt | tail -45
find $O/trace -name "*kernel_trace.csv" -delete
stamp "time lines"
