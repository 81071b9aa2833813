tag=${1:-r06g}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R
export LIO_SYNTH_CACHE=/tmp/lio_synth; mkdir -p $LIO_SYNTH_CACHE
T0=$(date +%s); stamp() { echo "== $1: $(( $(date +%s) - T0 )) s"; }
timeout 300 python scripts/dbg/map_update_steps.py 2>&1 | grep -a "calls us\|   fast" > $O/map_update_steps.log; cat $O/map_update_steps.log
stamp "map update steps"
(timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | grep -a "passed\|failed\|Error\|assert \|^E " | tail -12) > $O/pytest_gpu.log; cat $O/pytest_gpu.log
stamp pytest
