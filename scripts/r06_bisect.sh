R=$GRAFT_REPO_ROOT; cd $R
for v in "" head oldmembers; do
  if [ -n "$v" ]; then export LIO_HIP_LIBRARY=$R/better_fastlio2_amd/lib/ab/liblio_hip_$v.so; else unset LIO_HIP_LIBRARY; fi
  echo "== variant '$v'"
  timeout 300 python -m pytest tests/test_golden.py -m gpu -q -x 2>&1 | grep -a "passed\|failed\|assert " | tail -3
done
