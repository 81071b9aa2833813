# fuzz_loop with its round-6 twin (the node's own map_incremental call: no lists wanted, no host round trip) + the suite
tag=${1:-r06s2}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R
export LIO_SYNTH_CACHE=/tmp/lio_synth; mkdir -p $LIO_SYNTH_CACHE
timeout 1500 python tests/fuzz/fuzz_loop.py 400 97000 > $O/fuzz_loop.log 2>&1; echo "fuzz_loop rc=$? $(grep -a -o 'FUZZ[^\r]*' $O/fuzz_loop.log | tail -1 | cut -c1-300)"; grep -a "Error\|assert" $O/fuzz_loop.log | tail -5
(timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | grep -a "passed\|failed\|Error\|assert \|^E " | tail -12) > $O/pytest_gpu.log; cat $O/pytest_gpu.log
