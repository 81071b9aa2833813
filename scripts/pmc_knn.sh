cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_WAIT_ANY" "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_VMEM"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$tag -- python $R/scripts/knn_loop.py 8 6 > /dev/null 2>&1
done
python - <<'PY'
import glob, csv, collections, os
root=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out'
for f in sorted(glob.glob(root+'/pmc_*/**/*counter_collection.csv', recursive=True)):
    acc=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name']
        if 'knn27' in k or 'residual' in k or 'knn_tree' in k:
            acc[k[:60]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in acc.items():
        print(k, {c: (sum(x)/len(x), len(x)) for c,x in v.items()})
PY
