# A/B on ONE box: k-NN kernel of HEAD~ (lib/ab/liblio_hip_base.so, built by hand from `git show <rev>:.../lio_knn.hip`) against the
# working tree's -- the previous neighbours of a bounded search loaded beside the level-1 probes, the walk with eight candidates in flight.
#   gpurun --timeout 900 -- 'bash scripts/r05_knn_ab.sh r05c'
tag=${1:-r05c}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R
export LIO_SYNTH_CACHE=/tmp/lio_synth; mkdir -p $LIO_SYNTH_CACHE
BASE=$R/better_fastlio2_amd/lib/ab/liblio_hip_base.so
for rep in 1 2; do
  echo "== base (walk 4 only), repetition $rep"
  AB_LANES=8,16 AB_WALKS=4 AB_SIZES=100000,24000,8600 LIO_HIP_LIBRARY=$BASE timeout 300 python scripts/dbg/knn_walk_ab.py c2 2>&1 | grep -a "nq"
  echo "== new (walk 4 and 8), repetition $rep"
  AB_LANES=8,16 AB_WALKS=4,8 AB_SIZES=100000,24000,8600 timeout 300 python scripts/dbg/knn_walk_ab.py c2 2>&1 | grep -a "nq"
done > $O/knn_ab.log 2>&1
cat $O/knn_ab.log
echo "== c3"
(AB_LANES=8,16 AB_WALKS=4 LIO_HIP_LIBRARY=$BASE timeout 200 python scripts/dbg/knn_walk_ab.py c3 2>&1 | grep -a "nq" | sed 's/^/base /'; AB_LANES=8,16 AB_WALKS=4,8 timeout 200 python scripts/dbg/knn_walk_ab.py c3 2>&1 | grep -a "nq" | sed 's/^/new  /') > $O/knn_ab_c3.log 2>&1
cat $O/knn_ab_c3.log
timeout 300 python -m pytest tests/test_gpu_fullsize.py -q -x -k "c2 or c3" 2>&1 | tail -3
