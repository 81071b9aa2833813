# Round-3 evidence set on one box: gpurun --timeout 2400 -- 'bash scripts/end_of_round_r03.sh r03t'; then
# `python scripts/collect_profiles.py r03t` copies the summaries into profiles/ (see profiles/README.md).
tag=${1:-r03x}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R
bash scripts/end_of_round.sh $tag
# the driver's own command line
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_c2_driver.json 2> $O/bench_c2_driver.err; tail -1 $O/bench_c2_driver.json | cut -c1-200
# kernel time lines (one emulated-world step, one node-loop step, one full rebuild) + per-kernel totals of bench.py --steps 8
bash scripts/r03_trace.sh ${tag}_tl > $O/trace.out 2>&1; tail -3 $O/trace.out | cut -c1-200
for f in emulated_timeline.txt node_loop_timeline.txt rebuild_timeline.txt kernel_stats.txt; do cp $R/gpurun_out/${tag}_tl/$f $O/$f 2>/dev/null; done
bash scripts/rocprof_loopbody.sh ${tag}/prof_loop > $O/rocprof_loopbody.out 2>&1; tail -2 $O/rocprof_loopbody.out | cut -c1-200
LIO_TRACE_HOST=1 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-extras 2>&1 | grep "host trace" | tail -40 > $O/host_trace.log
bash scripts/r03_fuzz.sh $tag
