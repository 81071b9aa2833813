# End-of-round evidence set: run on the GPU box (gpurun --timeout 1500 -- 'bash scripts/end_of_round.sh r02f'), then
# `python scripts/collect_profiles.py r02f` copies the summaries into profiles/ (see profiles/README.md).
tag=${1:-r02x}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R
bash scripts/pmc_knn.sh > $O/pmc.out 2>&1; cp gpurun_out/pmc/knn_traffic.json $O/knn_traffic.json; cp $O/knn_traffic.json profiles/knn_traffic.json
timeout 900 python -m pytest tests -m gpu -q 2>&1 | grep -a "passed\|failed\|error" | tail -3 > $O/pytest_gpu.log; cat $O/pytest_gpu.log
timeout 400 python bench.py > $O/bench_c2.json 2> $O/bench_c2.err; tail -1 $O/bench_c2.json | cut -c1-260
timeout 300 python bench.py --config c3 > $O/bench_c3.json 2> $O/bench_c3.err; tail -1 $O/bench_c3.json | cut -c1-200
timeout 900 python bench.py --config c5 --steps 40 --warmup 5 --no-cpu-baseline > $O/bench_c5.json 2> $O/bench_c5.err; tail -1 $O/bench_c5.json | cut -c1-200
timeout 200 python scripts/loop_body_step.py 2>&1 | grep -a -v "thread\|amdgpu" > $O/loopbody.log; tail -7 $O/loopbody.log
timeout 200 python scripts/dbg/stamps.py 2>&1 | grep -a -v "thread\|amdgpu" > $O/stamps.log; head -4 $O/stamps.log
bash scripts/rocprof_bench.sh ${tag}/prof > $O/rocprof_bench.out 2>&1; tail -2 $O/rocprof_bench.out | cut -c1-200
