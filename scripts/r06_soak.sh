# Soak of the differential fuzzers on fresh seeds (final code of round 6): the map and loop fuzzers carry the round's changes
# (map update without its round trip, new key layout, sorts), the others re-run as a regression.
tag=${1:-r06s}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R
export LIO_SYNTH_CACHE=/tmp/lio_synth; mkdir -p $LIO_SYNTH_CACHE
run() { name=$1; shift; timeout 1200 python tests/fuzz/$name.py "$@" > $O/$name.log 2>&1; echo "$name rc=$? $(grep -a -o 'FUZZ[^\r]*' $O/$name.log | tail -1 | cut -c1-300)"; }
run fuzz_map 900 91000
run fuzz_loop 500 92000
run fuzz_measure 300 93000
run fuzz_scanprep 800 94000
run fuzz_pipeline 200 95000
run fuzz_assist 800 96000
