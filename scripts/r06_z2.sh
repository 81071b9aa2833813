R=$GRAFT_REPO_ROOT; cd $R
export LIO_SYNTH_CACHE=/tmp/lio_synth; mkdir -p $LIO_SYNTH_CACHE
echo "== round 6"; timeout 200 python scripts/dbg/node_loop_outliers.py 1500 2>&1 | grep -a "steps |\|slow call\|counters at"
echo "== round 5 build"; LIO_HIP_LIBRARY=$R/better_fastlio2_amd/lib/ab/liblio_hip_base.so timeout 200 python scripts/dbg/node_loop_outliers.py 1500 2>&1 | grep -a "steps |\|slow call\|counters at"
