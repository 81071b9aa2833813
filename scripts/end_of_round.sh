# End-of-round evidence set: run on the GPU box (gpurun -- 'bash scripts/end_of_round.sh r01p'), then copy
# gpurun_out/<tag>/* summaries into profiles/ (see profiles/README.md).
tag=${1:-r01p}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R
timeout 600 python -m pytest tests -m gpu -q 2>&1 | grep -a "passed\|failed\|error" | tail -3 > $O/pytest_gpu.log; cat $O/pytest_gpu.log
timeout 400 python bench.py > $O/bench_c2.log 2> $O/bench_c2.err; tail -1 $O/bench_c2.log | cut -c1-260
timeout 300 python bench.py --config c3 > $O/bench_c3.log 2> $O/bench_c3.err; tail -1 $O/bench_c3.log | cut -c1-200
timeout 300 python bench.py --no-cpu-baseline --map-update > $O/bench_c2_map_update.log 2> $O/bench_c2_map_update.err; tail -1 $O/bench_c2_map_update.log | cut -c1-200
timeout 200 python scripts/loop_body_step.py 2>&1 | grep -a -v "thread\|amdgpu" > $O/loopbody.log; tail -7 $O/loopbody.log
bash scripts/rocprof_bench.sh ${tag}_prof > $O/rocprof_bench.out 2>&1; tail -2 $O/rocprof_bench.out | cut -c1-200
