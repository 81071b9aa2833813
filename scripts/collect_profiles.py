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
                  ("knn_traffic.json", "knn_traffic.json"), ("knn_stamps.log", f"{tag}_knn_stamps.log"),
                  # round 3 (scripts/end_of_round_r03.sh)
                  ("bench_c2_driver.json", f"{tag}_bench_c2_driver_cmdline.json"), ("emulated_timeline.txt", f"{tag}_emulated_world8_timeline.txt"),
                  ("node_loop_timeline.txt", f"{tag}_node_loop_timeline.txt"), ("rebuild_timeline.txt", f"{tag}_rebuild_timeline.txt"),
                  ("kernel_stats.txt", f"{tag}_bench_c2_extras_kernel_totals.txt"), ("host_trace.log", f"{tag}_host_trace.log"), ("step_timeline.txt", f"{tag}_step_timeline.txt"),
                  # round 4 (scripts/end_of_round_r04.sh, scripts/lab/r04_timeline.sh)
                  ("emulated_timeline_deferred.txt", f"{tag}_emulated_world8_timeline_deferred.txt"),
                  ("emulated_timeline_immediate.txt", f"{tag}_emulated_world8_timeline_immediate.txt"),
                  (os.path.join("prof_loop", "kernel_stats_trim.csv"), f"{tag}_loopbody_kernel_stats.csv")):
    p = os.path.join(src, name)
    if os.path.exists(p) and os.path.getsize(p) > 0:
        shutil.copy(p, os.path.join(dst, out))
        print("copied", out)
ks = os.path.join(src, "prof", "kernel_stats.csv")
if os.path.exists(ks):
    subprocess.check_call([sys.executable, os.path.join(ROOT, "scripts", "trim_kernel_stats.py"), ks,
                           os.path.join(dst, f"{tag}_bench_c2_kernel_stats.csv"),
                           "rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extras (the CONTRACT LOOP only: the drive + its untimed event-stamped repeat; three of the first-search launches are passes armed across a scan boundary that lio_synchronize called off -- they stand down in ~5 us; scripts/rocprof_bench.sh)"])
    print("copied", f"{tag}_bench_c2_kernel_stats.csv")

for sub, name, out, cmd in (("prof", "kernel_stats_all.csv", f"{tag}_bench_c2_kernel_stats_all_extras.csv", "python bench.py --steps 100 --warmup 10 --no-cpu-baseline (EVERY extra: averages mix the drive, the node loop's denser map, the emulated world and the 2 400-step window)"),
                            ("prof_c3", "kernel_stats.csv", f"{tag}_bench_c3_kernel_stats.csv", "python bench.py --config c3 --steps 100 --warmup 10 --no-cpu-baseline --no-extras (the contract loop only)")):
    ks2 = os.path.join(src, sub, name)
    if os.path.exists(ks2):
        subprocess.check_call([sys.executable, os.path.join(ROOT, "scripts", "trim_kernel_stats.py"), ks2, os.path.join(dst, out),
                               "rocprofv3 --kernel-trace --stats --output-format csv -- " + cmd + " (scripts/rocprof_bench.sh)"])
        print("copied", out)

# the fuzzers' verdict lines
fz = [f for f in ("fuzz_map.log", "fuzz_loop.log", "fuzz_measure.log", "fuzz_scanprep.log", "fuzz_pipeline.log", "fuzz_assist.log") if os.path.exists(os.path.join(src, f))]
if fz:
    with open(os.path.join(dst, f"{tag}_fuzz.log"), "w") as out:
        for f in fz:
            # (the reference tree's rebuild thread prints into the same stream, without line ends: take the verdict itself)
            import re
            verdicts = re.findall(r"FUZZ[^\r\n]*", open(os.path.join(src, f), errors="replace").read())
            out.write(f"== {f}: " + (verdicts[-1] if verdicts else "no verdict line") + "\n")
    print("copied", f"{tag}_fuzz.log")
