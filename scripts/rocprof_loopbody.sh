# rocprofv3 kernel-trace of scripts/loop_body_step.py (the node's whole loop body); stats in gpurun_out/$1/
tag=${1:-r01o}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/$tag
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$tag/trace -- python $R/scripts/loop_body_step.py 20 > $R/gpurun_out/$tag/loop.log 2> $R/gpurun_out/$tag/loop.err
find $R/gpurun_out/$tag/trace -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/$tag/kernel_stats.csv \;
find $R/gpurun_out/$tag/trace -name "*kernel_trace.csv" -delete
python $R/scripts/trim_kernel_stats.py $R/gpurun_out/$tag/kernel_stats.csv $R/gpurun_out/$tag/kernel_stats_trim.csv "rocprofv3 --kernel-trace --stats -- python scripts/loop_body_step.py 20"
head -45 $R/gpurun_out/$tag/kernel_stats_trim.csv | cut -c1-150
grep -a -v "thread\|amdgpu" $R/gpurun_out/$tag/loop.log | tail -9
