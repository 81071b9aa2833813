# Final evidence set of round 4 on one box: gpurun --timeout 2400 -- 'bash scripts/end_of_round_r04.sh r04z'; then
# `python scripts/collect_profiles.py r04z` (+ the extra logs named in profiles/README.md).
tag=${1:-r04z}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R
export LIO_SYNTH_CACHE=/tmp/lio_synth
T0=$(date +%s); stamp() { echo "== $1: $(( $(date +%s) - T0 )) s"; }
bash scripts/pmc_knn.sh c2 c3 c5 > $O/pmc.out 2>&1; cp gpurun_out/pmc/knn_traffic.json $O/knn_traffic.json; cp $O/knn_traffic.json profiles/knn_traffic.json; grep -a "instructions per wavefront\|hbm_bytes" $O/pmc.out | cut -c1-200
stamp pmc
timeout 700 python -m pytest tests -m gpu -q 2>&1 | grep -a "passed\|failed\|error" | tail -3 > $O/pytest_gpu.log; cat $O/pytest_gpu.log
stamp pytest
timeout 500 python bench.py > $O/bench_c2.json 2> $O/bench_c2.err; tail -1 $O/bench_c2.json | cut -c1-200
stamp "bench c2"
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_c2_driver.json 2> $O/bench_c2_driver.err; tail -1 $O/bench_c2_driver.json | cut -c1-200
stamp "bench c2 driver cmdline"
timeout 400 python bench.py --config c3 > $O/bench_c3.json 2> $O/bench_c3.err; tail -1 $O/bench_c3.json | cut -c1-200
stamp "bench c3"
timeout 200 python scripts/loop_body_step.py 2>&1 | grep -a -v "thread\|amdgpu" > $O/loopbody.log; tail -7 $O/loopbody.log
stamp "loop body"
timeout 200 python scripts/dbg/stamps.py 2>&1 | grep -a -v "thread\|amdgpu" > $O/stamps.log; head -3 $O/stamps.log
bash scripts/rocprof_bench.sh ${tag}/prof > $O/rocprof_bench.out 2>&1; tail -2 $O/rocprof_bench.out | cut -c1-200
stamp "stamps + rocprof bench"
( cd /tmp && export TMPDIR=/tmp && LIO_BENCH_LONG_STEPS=0 timeout 500 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline > $O/trace_bench.log 2> $O/trace_bench.err )
python scripts/emulated_timeline.py $O/trace $O/emulated_timeline.txt | tail -40
find $O/trace -name "*kernel_trace.csv" -delete
stamp "emulated world time line"
timeout 200 python tests/fuzz/fuzz_map.py 120 5000 > $O/fuzz_map.log 2>&1; echo "fuzz_map rc=$?"
timeout 200 python tests/fuzz/fuzz_loop.py 80 9000 > $O/fuzz_loop.log 2>&1; echo "fuzz_loop rc=$?"
timeout 200 python tests/fuzz/fuzz_measure.py 150 12000 > $O/fuzz_measure.log 2>&1; echo "fuzz_measure rc=$?"
timeout 200 python tests/fuzz/fuzz_scanprep.py 200 40000 > $O/fuzz_scanprep.log 2>&1; echo "fuzz_scanprep rc=$?"
timeout 200 python tests/fuzz/fuzz_pipeline.py 100 64000 > $O/fuzz_pipeline.log 2>&1; echo "fuzz_pipeline rc=$?"
timeout 200 python tests/fuzz/fuzz_assist.py 400 74000 > $O/fuzz_assist.log 2>&1; echo "fuzz_assist rc=$?"
stamp fuzz
bash scripts/lab/r04_step_timeline.sh $tag > $O/step_timeline.out 2>&1; tail -3 $O/step_timeline.txt
LIO_TRACE_HOST=1 timeout 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-extras 2>&1 >/dev/null | grep -a "host trace" | head -90 | tail -48 > $O/host_trace.log
stamp "step time line + host trace"
timeout 1300 python bench.py --config c5 --steps 40 --warmup 5 > $O/bench_c5.json 2> $O/bench_c5.err; tail -1 $O/bench_c5.json | cut -c1-200
stamp "bench c5"
