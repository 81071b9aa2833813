# Round 6, second GPU visit: the map update without its host round trip (lio_map_incremental's fast path): suite, fuzz seeds,
# A/B against LIO_FAST_UPDATE=0 on ONE box (node loop of bench.py at C2 and C3, the node's loop body), kernel time line.
#   gpurun --timeout 1800 -- 'bash scripts/r06_b.sh r06b'
tag=${1:-r06b}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R
export LIO_SYNTH_CACHE=/tmp/lio_synth; mkdir -p $LIO_SYNTH_CACHE
T0=$(date +%s); stamp() { echo "== $1: $(( $(date +%s) - T0 )) s"; }
(timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | grep -a -v "thread\|amdgpu" | tail -15) > $O/pytest_gpu.log; tail -6 $O/pytest_gpu.log
stamp pytest
timeout 200 python tests/fuzz/fuzz_map.py 60 6500 > $O/fuzz_map.log 2>&1; echo "fuzz_map rc=$?"; tail -2 $O/fuzz_map.log
timeout 300 python tests/fuzz/fuzz_loop.py 60 10500 > $O/fuzz_loop.log 2>&1; echo "fuzz_loop rc=$?"; tail -2 $O/fuzz_loop.log
stamp fuzz
python - <<PY > $O/fast_update_ab.log 2>&1
import json, subprocess, sys, os
for cfg in ("c2", "c3"):
    for rep in range(3):
        for name, v in (("round trip", "0"), ("fast      ", "1")):
            env = dict(os.environ, LIO_FAST_UPDATE=v, LIO_BENCH_LONG_STEPS="0")
            out = subprocess.run([sys.executable, "bench.py", "--config", cfg, "--steps", "48", "--warmup", "5", "--no-cpu-baseline", "--long-steps", "0",
                                  "--other-configs", "", "--no-loop-body", "--emulate-world", "0"], env=env, capture_output=True, text=True).stdout.strip().splitlines()[-1]
            d = json.loads(out)
            nl = d["node_loop"]
            print(cfg, "rep", rep, name, "node_loop %.0f scans/s  step %.1f us  map update %.1f us  inserted %.0f  p50 %.1f" %
                  (nl["value"], nl["ms_per_step"] * 1e3, nl["map_update_ms_per_step"] * 1e3, nl["inserted_per_step"], nl["latency"]["p50"] * 1e3), flush=True)
PY
cat $O/fast_update_ab.log
stamp "fast update A/B"
for v in 0 1 0 1; do echo "LIO_FAST_UPDATE=$v"; LIO_FAST_UPDATE=$v timeout 300 python scripts/loop_body_step.py 40 2>&1 | grep -a "map_incremental\|SERIAL \|OVERLAPPED"; done > $O/loopbody_ab.log 2>&1
cat $O/loopbody_ab.log
stamp "loop body A/B"
for fu in 1 0; do ( cd /tmp && export TMPDIR=/tmp && LIO_FAST_UPDATE=$fu LIO_BENCH_LONG_STEPS=0 timeout 500 rocprofv3 --kernel-trace --output-format csv -d $O/trace$fu -- python $R/bench.py --steps 16 --warmup 2 --no-cpu-baseline --long-steps 0 --other-configs "" --no-loop-body --emulate-world 0 > $O/trace_bench$fu.log 2> $O/trace_bench$fu.err ); python scripts/node_loop_timeline.py $O/trace$fu $O/node_loop_timeline_fast$fu.txt; find $O/trace$fu -name "*kernel_trace.csv" -delete; done
stamp "node loop time line"
