# round 4: assisted pull -- targeted tests, then the in-process A/B on the callback chain (8 clouds in turn = cold; 1 cloud = warm in L3)
tag=${1:-r04y}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R; export LIO_SYNTH_CACHE=/tmp/lio_synth
timeout 600 python -m pytest tests/test_gpu_callback.py -m gpu -q -x 2>&1 | grep -a -v "Rebuild thread\|Multi thread\|amdgpu.ids" | tail -12
timeout 400 python scripts/dbg/assist_pull_ab.py c2 2 200 0 1 70 2>&1 | grep -a -v "Rebuild thread\|Multi thread\|amdgpu.ids" | tee $O/assist_ab_c2.log
ROTATE=1 timeout 400 python scripts/dbg/assist_pull_ab.py c2 2 200 0 1 85 2>&1 | grep -a -v "Rebuild thread\|Multi thread\|amdgpu.ids" | tee $O/assist_ab_c2_one_cloud.log
timeout 300 python scripts/dbg/assist_pull_ab.py c3 3 200 0 1 50 70 2>&1 | grep -a -v "Rebuild thread\|Multi thread\|amdgpu.ids" | tee $O/assist_ab_c3.log
