# round 4: tests + one bench line (driver form) into gpurun_out/$1
tag=${1:-r04a}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R; export LIO_SYNTH_CACHE=/tmp/lio_synth
T0=$(date +%s); stamp() { echo "== $1: $(( $(date +%s) - T0 )) s"; }
timeout 900 python -m pytest tests -m gpu -q -x ${PYTEST_ARGS:-} 2>&1 | grep -a -v "Rebuild thread\|Multi thread\|amdgpu.ids" > $O/pytest_gpu_full.log; grep -a "passed\|failed\|error" $O/pytest_gpu_full.log | tail -3 > $O/pytest_gpu.log; tail -${PYTEST_TAIL:-25} $O/pytest_gpu_full.log
stamp pytest
if [ "${2:-bench}" = "bench" ]; then
timeout 500 python bench.py --gpus 1 --steps 100 --warmup 10 > $O/bench_c2.json 2> $O/bench_c2.err; tail -c 600 $O/bench_c2.err
python - <<PY
import json
j=json.loads(open("$O/bench_c2.json").read().strip().splitlines()[-1])
print("value", j["value"], "ms", j["ms_per_step"], "lat", j["latency"])
print("kernels", j["kernels_us"], "frac", j["roofline"]["frac"], j["roofline"]["avg_launch_us"])
for k in ("value_resident","value_serial_upload","value_same_scan_replay","value_callback","value_callback_patched"):
    v=j.get(k); print(k, v and {a:v[a] for a in v if a in ("value","ms_per_step","ms_per_step_inside_h_share_model","ms_per_step_filter_algebra_on_the_rows")})
n=j.get("callback_at_node_scan_size")
if n:
    for k,v in n.items():
        print("node", k, v if not isinstance(v,dict) else {a:v[a] for a in v if a in ("value","ms_per_step","ms_per_step_inside_h_share_model")})
print("node_loop", {a:j["node_loop"][a] for a in ("value","ms_per_step","map_update_ms_per_step")})
e=j.get("node_loop_emulated_world8"); print("world8", e and {a:e.get(a) for a in ("value","ms_per_step","scan_step_ms","map_update_ms_per_step","vs_node_loop_world1")})
print("parity", {k:v for k,v in j.get("parity",{}).items() if k in ("pos_diff_m","rot_diff_rad")}, (j.get("parity",{}).get("neighbors_bit_exact") or {}))
PY
stamp bench
fi
