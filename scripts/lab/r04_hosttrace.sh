tag=${1:-r04s}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R; export LIO_SYNTH_CACHE=/tmp/lio_synth
timeout 600 python -m pytest tests -m gpu -q -x -k "scan_boundary and not_armed" 2>&1 | grep -a -v "Rebuild thread\|Multi thread\|amdgpu.ids" | tail -5
for v in "0 3" "1 3" "1 7"; do set -- $v
LIO_TRACE_HOST=1 LIO_ARM_NEXT_SCAN=$1 LIO_GATE_IN_KERNEL=$2 timeout 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-extras 2>&1 >/dev/null | grep -a "host trace" | head -90 > $O/host_trace_$1_$2.log
echo "== arm $1 gik $2"; tail -48 $O/host_trace_$1_$2.log
done
