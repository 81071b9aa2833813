tag=${1:-r04y5}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R; export LIO_SYNTH_CACHE=/tmp/lio_synth
timeout 600 python -m pytest tests/test_gpu_callback.py -m gpu -q -x 2>&1 | grep -a -v "Rebuild thread\|Multi thread\|amdgpu.ids" | tail -5
timeout 400 python scripts/dbg/assist_pull_ab.py c2 3 150 0 1 80 2>&1 | grep -a -v "Rebuild thread\|Multi thread\|amdgpu.ids" | tee -a $O/assist_rot8.log
ROTATE=1 timeout 400 python scripts/dbg/assist_pull_ab.py c2 3 150 0 1 80 2>&1 | grep -a -v "Rebuild thread\|Multi thread\|amdgpu.ids" | tee -a $O/assist_rot1.log
timeout 300 python scripts/dbg/assist_pull_ab.py c3 3 150 0 1 80 2>&1 | grep -a -v "Rebuild thread\|Multi thread\|amdgpu.ids" | tee $O/assist_c3.log
