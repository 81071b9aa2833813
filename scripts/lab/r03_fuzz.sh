# Round-3 differential fuzzers on the GPU box (new map-maintenance kernels): exit codes + tails
#   gpurun --timeout 1500 -- 'bash scripts/r03_fuzz.sh r03k'
tag=${1:-r03x}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R
timeout 400 python tests/fuzz/fuzz_map.py 250 5000 > $O/fuzz_map.log 2>&1; echo "fuzz_map rc=$?"; grep -a -v thread $O/fuzz_map.log | tail -2
timeout 400 python tests/fuzz/fuzz_loop.py 120 9000 > $O/fuzz_loop.log 2>&1; echo "fuzz_loop rc=$?"; grep -a -v thread $O/fuzz_loop.log | tail -2
timeout 300 python tests/fuzz/fuzz_measure.py 200 12000 > $O/fuzz_measure.log 2>&1; echo "fuzz_measure rc=$?"; grep -a -v thread $O/fuzz_measure.log | tail -2
timeout 300 python tests/fuzz/fuzz_scanprep.py 200 40000 > $O/fuzz_scanprep.log 2>&1; echo "fuzz_scanprep rc=$?"; grep -a -v thread $O/fuzz_scanprep.log | tail -2
