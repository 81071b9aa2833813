"""Shortens a rocprofv3 `*_kernel_stats.csv` (template-heavy rocPRIM names) into the form kept under profiles/.

    python scripts/trim_kernel_stats.py gpurun_out/r01j/kernel_stats.csv profiles/r01j_x_kernel_stats.csv "command line"
"""
import csv
import re
import sys


def short(n: str) -> str:
    n = re.sub(r"rocprim::ROCPRIM_\d+_NS::", "rocprim::", n)
    if n.startswith("void rocprim"):
        m = re.findall(r"rocprim::detail::(radix_sort_onesweep\w*|\w+_kernel\w*|lookback_scan\w*|partition\w*|merge\w*|scan\w*|"
                       r"reduce_by_key\w*|\w+_impl\w*)", n)
        return "rocprim::" + (m[1] if len(m) > 1 else (m[0] if m else "kernel"))
    return n if len(n) < 120 else n[:117] + "..."


def main():
    src, dst = sys.argv[1], sys.argv[2]
    note = sys.argv[3] if len(sys.argv) > 3 else ""
    rows = list(csv.DictReader(open(src)))
    with open(dst, "w") as f:
        f.write('"# %s"\n' % note.replace('"', "'"))
        f.write("Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs\n")
        for r in rows[:40]:
            f.write('"%s",%s,%s,%s,%s,%s,%s\n' % (short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"],
                                                 r["Percentage"], r["MinNs"], r["MaxNs"]))


if __name__ == "__main__":
    main()
