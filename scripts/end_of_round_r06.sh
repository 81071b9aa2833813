# Final evidence set of round 6 on one box: gpurun --timeout 2700 -- 'bash scripts/end_of_round_r06.sh r06z'; then
# `python scripts/collect_profiles.py r06z` (+ the extra logs named in profiles/README.md).
tag=${1:-r06z}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R
export LIO_SYNTH_CACHE=/tmp/lio_synth
T0=$(date +%s); stamp() { echo "== $1: $(( $(date +%s) - T0 )) s"; }
bash scripts/pmc_knn.sh c2 c3 c5 > $O/pmc.out 2>&1; cp gpurun_out/pmc/knn_traffic.json $O/knn_traffic.json; cp $O/knn_traffic.json profiles/knn_traffic.json; grep -a "instructions per wavefront\|hbm_bytes" $O/pmc.out | cut -c1-200
stamp pmc
(timeout 900 python -m pytest tests -m gpu -q 2>&1 | grep -a "passed\|failed\|error" | tail -3) > $O/pytest_gpu.log; cat $O/pytest_gpu.log
stamp pytest
timeout 500 python bench.py > $O/bench_c2.json 2> $O/bench_c2.err; tail -c 2200 $O/bench_c2.json
stamp "bench c2"
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_c2_driver.json 2> $O/bench_c2_driver.err; tail -c 2200 $O/bench_c2_driver.json
stamp "bench c2 driver cmdline"
timeout 400 python bench.py --config c3 > $O/bench_c3.json 2> $O/bench_c3.err; tail -c 1200 $O/bench_c3.json
stamp "bench c3"
timeout 300 python scripts/loop_body_step.py 60 2>&1 | grep -a -v "thread\|amdgpu" > $O/loopbody.log; tail -10 $O/loopbody.log
stamp "loop body"
timeout 200 python scripts/dbg/stamps.py 2>&1 | grep -a -v "thread\|amdgpu" > $O/stamps.log; head -3 $O/stamps.log
bash scripts/rocprof_bench.sh ${tag}/prof > $O/rocprof_bench.out 2>&1; tail -2 $O/rocprof_bench.out | cut -c1-200
stamp "stamps + rocprof bench"
( cd /tmp && export TMPDIR=/tmp && LIO_BENCH_LONG_STEPS=0 timeout 500 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --long-steps 0 --other-configs "" --no-loop-body > $O/trace_bench.log 2> $O/trace_bench.err )
python scripts/emulated_timeline.py $O/trace $O/emulated_timeline_deferred.txt | tail -30
python scripts/emulated_timeline.py $O/trace $O/emulated_timeline_immediate.txt 17 > /dev/null
find $O/trace -name "*kernel_trace.csv" -delete
( cd /tmp && export TMPDIR=/tmp && LIO_BENCH_LONG_STEPS=0 timeout 500 rocprofv3 --kernel-trace --output-format csv -d $O/trace1 -- python $R/bench.py --steps 16 --warmup 2 --no-cpu-baseline --long-steps 0 --other-configs "" --no-loop-body --emulate-world 0 > $O/trace1_bench.log 2> $O/trace1_bench.err )
python scripts/node_loop_timeline.py $O/trace1 $O/node_loop_timeline.txt | tail -18
find $O/trace1 -name "*kernel_trace.csv" -delete
stamp "time lines"
timeout 200 python tests/fuzz/fuzz_map.py 120 15500 > $O/fuzz_map.log 2>&1; echo "fuzz_map rc=$?"
timeout 200 python tests/fuzz/fuzz_loop.py 80 19500 > $O/fuzz_loop.log 2>&1; echo "fuzz_loop rc=$?"
timeout 200 python tests/fuzz/fuzz_measure.py 150 22500 > $O/fuzz_measure.log 2>&1; echo "fuzz_measure rc=$?"
timeout 200 python tests/fuzz/fuzz_scanprep.py 200 50500 > $O/fuzz_scanprep.log 2>&1; echo "fuzz_scanprep rc=$?"
timeout 200 python tests/fuzz/fuzz_pipeline.py 100 74500 > $O/fuzz_pipeline.log 2>&1; echo "fuzz_pipeline rc=$?"
timeout 200 python tests/fuzz/fuzz_assist.py 400 84500 > $O/fuzz_assist.log 2>&1; echo "fuzz_assist rc=$?"
stamp fuzz
timeout 1300 python bench.py --config c5 --steps 40 --warmup 5 > $O/bench_c5.json 2> $O/bench_c5.err; tail -c 1500 $O/bench_c5.json
stamp "bench c5"
