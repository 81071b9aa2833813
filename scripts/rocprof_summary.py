"""Turns a rocprofv3 (rocpd sqlite) result into the short per-kernel summary kept under profiles/.

    python scripts/rocprof_summary.py gpurun_out/prof_r01/bench_results.db profiles/r01_bench_c2_kernel_stats.md "command line"
"""
import re
import sqlite3
import sys


def short(name: str) -> str:
    name = re.sub(r"rocprim::ROCPRIM_\d+_NS::", "rocprim::", name)
    m = re.search(r"rocprim::detail::(\w+)", name)
    if name.startswith("void rocprim") and m:
        inner = re.findall(r"rocprim::detail::(\w+_impl\w*|radix_sort_onesweep\w*|\w+_kernel\w*|lookback_scan\w*|partition\w*|scan\w*)", name)
        return "rocprim::" + (inner[1] if len(inner) > 1 else m.group(1))
    return name if len(name) < 110 else name[:107] + "..."


def main():
    db, out = sys.argv[1], sys.argv[2]
    cmd = sys.argv[3] if len(sys.argv) > 3 else ""
    cur = sqlite3.connect(db).cursor()
    rows = list(cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    with open(out, "w") as f:
        f.write(f"# rocprofv3 --kernel-trace --stats summary\n\ncommand: `{cmd}`\n\n")
        f.write("| kernel | calls | total us | avg us | % |\n|---|---:|---:|---:|---:|\n")
        for name, calls, tot, avg, pct in rows[:24]:
            f.write(f"| `{short(name)}` | {calls} | {tot:.1f} | {avg:.3f} | {pct:.2f} |\n")
    print(open(out).read())


if __name__ == "__main__":
    main()
