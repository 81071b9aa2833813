# the two bench lines of the final evidence set again (bench.py's `summary` was re-ordered after scripts/end_of_round_r06.sh ran)
tag=${1:-r06z}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R
export LIO_SYNTH_CACHE=/tmp/lio_synth; mkdir -p $LIO_SYNTH_CACHE
timeout 500 python bench.py > $O/bench_c2.json 2> $O/bench_c2.err; tail -c 2100 $O/bench_c2.json; echo
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_c2_driver.json 2> $O/bench_c2_driver.err; tail -c 2100 $O/bench_c2_driver.json; echo
timeout 400 python bench.py --config c3 > $O/bench_c3.json 2> $O/bench_c3.err; tail -c 600 $O/bench_c3.json; echo
