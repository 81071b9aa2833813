"""Copies the summaries of an end-of-round GPU run (gpurun_out/<tag>/, scripts/end_of_round.sh) into profiles/."""
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
src = os.path.join(ROOT, "gpurun_out", tag)
dst = os.path.join(ROOT, "profiles")
for name, out in (("bench_c2.json", f"{tag}_bench_c2.json"), ("bench_c3.json", f"{tag}_bench_c3.json"), ("bench_c5.json", f"{tag}_bench_c5.json"),
                  ("loopbody.log", f"{tag}_loopbody.log"), ("pytest_gpu.log", f"{tag}_pytest_gpu.log"), ("stamps.log", f"{tag}_stamps.log"),
                  ("knn_traffic.json", "knn_traffic.json")):
    p = os.path.join(src, name)
    if os.path.exists(p) and os.path.getsize(p) > 0:
        shutil.copy(p, os.path.join(dst, out))
        print("copied", out)
ks = os.path.join(src, "prof", "kernel_stats.csv")
if os.path.exists(ks):
    subprocess.check_call([sys.executable, os.path.join(ROOT, "scripts", "trim_kernel_stats.py"), ks,
                           os.path.join(dst, f"{tag}_bench_c2_kernel_stats.csv"),
                           "rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --steps 100 --warmup 10 --no-cpu-baseline (scripts/rocprof_bench.sh)"])
    print("copied", f"{tag}_bench_c2_kernel_stats.csv")
