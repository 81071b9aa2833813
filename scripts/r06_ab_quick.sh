# quick one-box check of a change to the rebuild by cell runs: map parity tests + the forced-rebuild timing against the point sort
tag=${1:-r07m}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R
export LIO_SYNTH_CACHE=/tmp/lio_synth; mkdir -p $LIO_SYNTH_CACHE
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "rebuild or index or map or reconstruct or add_points or incremental" > $O/pytest.log 2>&1; grep -a -E "passed|failed|Error" $O/pytest.log | tail -3
for v in 1 0 1; do LIO_REBUILD_BY_CELLS=$v timeout 300 python scripts/rebuild_ab.py 2>&1 | tail -1 | tee -a $O/rebuild_ab.log; done
