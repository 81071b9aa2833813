# round 4: second chance inside knn27 -- tests (PYTEST_ARGS narrows), stamps, bench c2 (driver form) + c3, rocprofv3 kernel stats
tag=${1:-r04c}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R; export LIO_SYNTH_CACHE=/tmp/lio_synth
PYTEST_TAIL=30 bash scripts/r04_check.sh $tag nobench
timeout 200 python scripts/dbg/stamps.py 2>&1 | grep -a "us since" > $O/stamps.log; cat $O/stamps.log
for tune in 2 0; do
  LIO_KNN_TUNE=$tune timeout 300 python bench.py --gpus 1 --steps 60 --warmup 10 --no-cpu-baseline --no-extras > $O/bench_c2_tune$tune.json 2> $O/bench_c2_tune$tune.err
  LIO_KNN_TUNE=$tune timeout 300 python bench.py --config c3 --steps 100 --warmup 10 --no-cpu-baseline --no-extras > $O/bench_c3_tune$tune.json 2> $O/bench_c3_tune$tune.err
done
python - <<PY
import json
for cfg in ("c2","c3"):
    for t in (2,0):
        try:
            j=json.loads(open("$O/bench_%s_tune%d.json"%(cfg,t)).read().strip().splitlines()[-1])
            print(cfg,"tune",t,"value %.0f ms %.4f"%(j["value"],j["ms_per_step"]),"kernels",{k:round(v,2) for k,v in j["kernels_us"].items()},"unres",j["config"]["unresolved_queries"],"frac %.3f"%j["roofline"]["frac"])
        except Exception as e: print(cfg,t,"failed",e)
PY
