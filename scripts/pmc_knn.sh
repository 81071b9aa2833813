# PMC counters of the k-NN kernel (its own runs, --pmc with --kernel-trace only): instruction mix and the memory-side
# traffic behind bench.py's roofline.traffic.  Writes gpurun_out/pmc/knn_traffic.json (copy it to profiles/).
#   gpurun --timeout 900 -- 'bash scripts/pmc_knn.sh [c2|c3|c5 ...]'      (default: c2; every config named gets its entry)
cd /tmp && export TMPDIR=/tmp
export LIO_SYNTH_CACHE=/tmp/lio_synth        # the four passes share the generated world
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/pmc; mkdir -p $R/gpurun_out/pmc
cp $R/profiles/knn_traffic.json $R/gpurun_out/pmc/knn_traffic.json 2>/dev/null
for cfg in ${@:-c2}; do
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 400 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmc/$cfg/set$i -- python $R/scripts/knn_loop.py 0 6 $cfg > /dev/null 2>&1
done
python $R/scripts/pmc_traffic.py $R/gpurun_out/pmc $cfg
done
