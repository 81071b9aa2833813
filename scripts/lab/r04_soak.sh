# round 4: soak of the four differential fuzzers on fresh seeds (the final kernels), verdict lines into gpurun_out/$1/soak.log
tag=${1:-r04w}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R; export LIO_SYNTH_CACHE=/tmp/lio_synth
run() { name=$1; shift; timeout 190 python tests/fuzz/$name.py "$@" > $O/$name.log 2>&1; rc=$?; echo "== $name $* rc=$rc: $(grep -a -o "FUZZ[^\r]*" $O/$name.log | tail -2 | tr '\n' ' ')" | tee -a $O/soak.log; }
run fuzz_map 300 22000
run fuzz_loop 150 32000
run fuzz_measure 300 42000
run fuzz_scanprep 300 52000
run fuzz_pipeline 200 65000
run fuzz_assist 600 75000
