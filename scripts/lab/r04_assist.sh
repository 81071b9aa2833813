# round 4: assisted pull -- the callback tests, then the in-process A/B on the callback chain (8 clouds in turn / one cloud; one and two packing threads)
tag=${1:-r04yb}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R; export LIO_SYNTH_CACHE=/tmp/lio_synth
timeout 600 python -m pytest tests/test_gpu_callback.py -m gpu -q -x 2>&1 | grep -a -v "Rebuild thread\|Multi thread\|amdgpu.ids" | tail -5
for th in 1 2; do
THREADS=$th timeout 400 python scripts/dbg/assist_pull_ab.py c2 3 150 0 1 85 2>&1 | grep -a -v "Rebuild thread\|Multi thread\|amdgpu.ids" | tee -a $O/assist_rot8.log
THREADS=$th ROTATE=1 timeout 400 python scripts/dbg/assist_pull_ab.py c2 3 150 1 90 2>&1 | grep -a -v "Rebuild thread\|Multi thread\|amdgpu.ids" | tee -a $O/assist_rot1.log
done
