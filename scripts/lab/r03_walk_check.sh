# Round-3 GPU visit for the candidate-walk variants of knn27_kernel: bit-identity tests, A/B timings over scan sizes, the
# idle / cold probe, and the whole -m gpu suite with the four-in-flight walk forced on every 8- and 16-lane launch.
#   gpurun --timeout 900 -- 'bash scripts/r03_walk_check.sh r03y'
tag=${1:-r03y}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R
export LIO_SYNTH_CACHE=/tmp/lio_synth
T0=$(date +%s)
stamp() { echo "== $1: $(( $(date +%s) - T0 )) s since start"; }
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "candidate_walk or nearest_search_bit_exact" 2>&1 | grep -a -E "passed|failed|error|Error|assert" | tail -8
stamp "walk tests"
timeout 300 python scripts/dbg/knn_walk_ab.py c2 2>&1 | grep -a "nq" | tee $O/walk_ab_c2.log
stamp "A/B c2"
timeout 200 python scripts/dbg/knn_walk_ab.py c3 2>&1 | grep -a "nq" | tee $O/walk_ab_c3.log
stamp "A/B c3"
timeout 300 python scripts/dbg/idle_probe.py 2>&1 | grep -a "mode" | tee $O/idle_probe.log
stamp "idle probe"
(LIO_KNN_WALK=4 timeout 600 python -m pytest tests -m gpu -q 2>&1 | grep -a -E "passed|failed|error|^FAILED|^ERROR" | tail -12) | tee $O/pytest_gpu_walk4.log
stamp "pytest -m gpu with LIO_KNN_WALK=4"
