# intermediate check of the rebuild by cell runs + fused scan preparation: whole GPU suite, fuzzers that rebuild, A/B, default bench line
tag=${1:-r07d}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R
export LIO_SYNTH_CACHE=/tmp/lio_synth; mkdir -p $LIO_SYNTH_CACHE
(timeout 1200 python -m pytest tests -m gpu -q 2>&1 | grep -a "passed\|failed\|rror" | tail -5) > $O/pytest_gpu.log; cat $O/pytest_gpu.log
for v in 1 0 1 0; do LIO_REBUILD_BY_CELLS=$v timeout 300 python scripts/rebuild_ab.py 2>&1 | tail -1 | tee -a $O/rebuild_ab.log; done
timeout 300 python tests/fuzz/fuzz_map.py 150 16500 > $O/fuzz_map.log 2>&1; echo "fuzz_map rc=$?"; tail -1 $O/fuzz_map.log | cut -c1-300
timeout 300 python tests/fuzz/fuzz_loop.py 100 20500 > $O/fuzz_loop.log 2>&1; echo "fuzz_loop rc=$?"; tail -1 $O/fuzz_loop.log | cut -c1-300
timeout 300 python tests/fuzz/fuzz_pipeline.py 100 75500 > $O/fuzz_pipeline.log 2>&1; echo "fuzz_pipeline rc=$?"; tail -1 $O/fuzz_pipeline.log | cut -c1-300
timeout 500 python bench.py > $O/bench_c2.json 2> $O/bench_c2.err; tail -c 2300 $O/bench_c2.json
