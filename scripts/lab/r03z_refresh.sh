# refresh of the r03z set after the last changes (walk 4 at every size, bench notes): tests, bench lines c2 / driver form / c3 / c5
tag=r03z
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R; export LIO_SYNTH_CACHE=/tmp/lio_synth
T0=$(date +%s); stamp() { echo "== $1: $(( $(date +%s) - T0 )) s"; }
timeout 600 python -m pytest tests -m gpu -q 2>&1 | grep -a "passed\|failed\|error" | tail -3 > $O/pytest_gpu.log; cat $O/pytest_gpu.log
stamp pytest
timeout 400 python bench.py > $O/bench_c2.json 2> $O/bench_c2.err; tail -1 $O/bench_c2.json | cut -c1-160
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_c2_driver.json 2> $O/bench_c2_driver.err; tail -1 $O/bench_c2_driver.json | cut -c1-160
timeout 300 python bench.py --config c3 > $O/bench_c3.json 2> $O/bench_c3.err; tail -1 $O/bench_c3.json | cut -c1-160
stamp "bench c2 c3"
timeout 400 python bench.py --config c5 --steps 40 --warmup 5 --no-cpu-baseline > $O/bench_c5.json 2> $O/bench_c5.err; tail -1 $O/bench_c5.json | cut -c1-160
stamp "bench c5"
timeout 200 python scripts/loop_body_step.py 2>&1 | grep -a -v "thread\|amdgpu" > $O/loopbody.log; tail -7 $O/loopbody.log
