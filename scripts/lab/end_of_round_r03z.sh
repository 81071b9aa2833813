# Final evidence set of round 3 (tag r03z) on one box: gpurun --timeout 1200 -- 'bash scripts/end_of_round_r03z.sh r03z'; then
# `python scripts/collect_profiles.py r03z` (+ the extra logs named in profiles/README.md).
tag=${1:-r03z}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R
export LIO_SYNTH_CACHE=/tmp/lio_synth
T0=$(date +%s); stamp() { echo "== $1: $(( $(date +%s) - T0 )) s"; }
bash scripts/pmc_knn.sh > $O/pmc.out 2>&1; cp gpurun_out/pmc/knn_traffic.json $O/knn_traffic.json; cp $O/knn_traffic.json profiles/knn_traffic.json; tail -3 $O/pmc.out | cut -c1-200
stamp pmc
timeout 600 python -m pytest tests -m gpu -q 2>&1 | grep -a "passed\|failed\|error" | tail -3 > $O/pytest_gpu.log; cat $O/pytest_gpu.log
stamp pytest
timeout 400 python bench.py > $O/bench_c2.json 2> $O/bench_c2.err; tail -1 $O/bench_c2.json | cut -c1-200
stamp "bench c2"
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_c2_driver.json 2> $O/bench_c2_driver.err; tail -1 $O/bench_c2_driver.json | cut -c1-200
stamp "bench c2 driver cmdline"
timeout 300 python bench.py --config c3 > $O/bench_c3.json 2> $O/bench_c3.err; tail -1 $O/bench_c3.json | cut -c1-200
stamp "bench c3"
timeout 200 python scripts/loop_body_step.py 2>&1 | grep -a -v "thread\|amdgpu" > $O/loopbody.log; tail -7 $O/loopbody.log
bash scripts/rocprof_loopbody.sh ${tag}/prof_loop > $O/rocprof_loopbody.out 2>&1
bash scripts/dbg/voxel_timeline.sh > $O/voxel_timeline.txt 2>&1; cat $O/voxel_timeline.txt | tail -14
stamp "loop body"
timeout 200 python scripts/dbg/stamps.py 2>&1 | grep -a -v "thread\|amdgpu" > $O/stamps.log; head -3 $O/stamps.log
bash scripts/rocprof_bench.sh ${tag}/prof > $O/rocprof_bench.out 2>&1; tail -2 $O/rocprof_bench.out | cut -c1-200
stamp "stamps + rocprof bench"
(timeout 200 python scripts/dbg/knn_walk_ab.py c2; timeout 100 python scripts/dbg/knn_walk_ab.py c3) 2>&1 | grep -a "nq" > $O/walk_ab.log; tail -3 $O/walk_ab.log
timeout 100 python scripts/dbg/post_update_probe.py 2>&1 | grep -a "first" > $O/post_update_probe.log; cat $O/post_update_probe.log | cut -c1-200
timeout 100 python scripts/dbg/idle_probe.py 2>&1 | grep -a "mode" > $O/idle_probe.log
stamp "probes"
timeout 200 python tests/fuzz/fuzz_map.py 120 5000 > $O/fuzz_map.log 2>&1; echo "fuzz_map rc=$?"
timeout 200 python tests/fuzz/fuzz_loop.py 80 9000 > $O/fuzz_loop.log 2>&1; echo "fuzz_loop rc=$?"
timeout 200 python tests/fuzz/fuzz_measure.py 150 12000 > $O/fuzz_measure.log 2>&1; echo "fuzz_measure rc=$?"
timeout 200 python tests/fuzz/fuzz_scanprep.py 200 40000 > $O/fuzz_scanprep.log 2>&1; echo "fuzz_scanprep rc=$?"
stamp fuzz
timeout 400 python bench.py --config c5 --steps 40 --warmup 5 --no-cpu-baseline > $O/bench_c5.json 2> $O/bench_c5.err; tail -1 $O/bench_c5.json | cut -c1-200
stamp "bench c5"
