# Soak of the six differential fuzzers on fresh seeds (final code of round 5).
tag=${1:-r05s}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R
export LIO_SYNTH_CACHE=/tmp/lio_synth; mkdir -p $LIO_SYNTH_CACHE
run() { name=$1; shift; timeout 900 python tests/fuzz/$name.py "$@" > $O/$name.log 2>&1; echo "$name rc=$? $(grep -a -o 'FUZZ[^\r]*' $O/$name.log | tail -1 | cut -c1-300)"; }
run fuzz_map 600 81000
run fuzz_loop 320 82000
run fuzz_measure 600 83000
run fuzz_scanprep 1500 84000
run fuzz_pipeline 400 85000
run fuzz_assist 1600 86000
