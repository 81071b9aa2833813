# Round 5 lab visit: the -m gpu suite on the split library, the robustness tests, the overlapped loop body, the k-NN stamp ladder.
#   gpurun --timeout 1500 -- 'bash scripts/r05_lab.sh r05b'
tag=${1:-r05b}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R
export LIO_SYNTH_CACHE=/tmp/lio_synth; mkdir -p $LIO_SYNTH_CACHE
T0=$(date +%s); stamp() { echo "== $1: $(( $(date +%s) - T0 )) s"; }
(timeout 1000 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_fullsize.py::test_c5_sampled_neighbours_equal_the_reference_tree --deselect tests/test_gpu_fullsize.py::test_c5_properties_1m_scan_50m_map 2>&1 | grep -a -v "thread\|amdgpu" | tail -40) > $O/pytest_gpu.log; tail -12 $O/pytest_gpu.log
stamp pytest
timeout 300 python scripts/loop_body_step.py 2>&1 | grep -a -v "thread\|amdgpu" > $O/loopbody.log; tail -12 $O/loopbody.log
stamp "loop body"
timeout 300 python scripts/knn_stamps.py c3 c2 2>&1 | grep -a -v "thread\|amdgpu" > $O/knn_stamps.log; cat $O/knn_stamps.log
stamp "knn stamps"
