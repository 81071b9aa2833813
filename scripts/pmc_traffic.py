"""Reduces the rocprofv3 --pmc passes of scripts/pmc_knn.sh to profiles/knn_traffic.json: per-launch averages of the
counters of knn27_kernel (first search and bounded re-search), stamped with the hash of the kernel's sources -- bench.py
reports roofline.traffic only while that hash matches the tree it runs from.

HBM-side bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE
reports half of the bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM section); WRITE_SIZE is uncalibrated.
"""
import collections
import re
import csv
import datetime
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (knn_source_hash)


def main():
    root = sys.argv[1]
    cfg = sys.argv[2] if len(sys.argv) > 2 else "c2"
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    shapes = collections.defaultdict(set)
    for f in sorted(glob.glob(os.path.join(root, cfg, "**", "*counter_collection.csv"), recursive=True)):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "knn27_kernel" not in k:
                continue
            # knn27_kernel<G, CENSUS, HINT[, U]>: the third template argument says which search it is
            m = re.search(r"knn27_kernel<(\d+),(true|false),(true|false)(?:,(\d+))?>", k.replace(" ", ""))
            variant = "hinted_search" if m and m.group(3) == "true" else "first_search"
            shapes[variant].add(k.replace(" ", "").split("(")[0].split("lio::")[-1])
            acc[variant][r["Counter_Name"]].append(float(r["Counter_Value"]))
    out = {}
    for variant, counters in acc.items():
        avg = {c: sum(v) / len(v) for c, v in counters.items()}
        n = {c: len(v) for c, v in counters.items()}
        e = {"launches_averaged": min(n.values()) if n else 0, "counters_per_launch": avg}
        if "FETCH_SIZE" in avg and "WRITE_SIZE" in avg:
            e["FETCH_SIZE_KB"] = avg["FETCH_SIZE"]; e["WRITE_SIZE_KB"] = avg["WRITE_SIZE"]
            e["hbm_bytes_per_launch"] = (2.0 * avg["FETCH_SIZE"] + avg["WRITE_SIZE"]) * 1024.0
        if "SQ_WAVES" in avg and avg["SQ_WAVES"] > 0:
            e["instructions_per_wavefront"] = {c: avg[c] / avg["SQ_WAVES"] for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS") if c in avg}
        out[variant] = e
    first = out.get("first_search", {})
    path = os.path.join(root, "knn_traffic.json")
    doc = json.load(open(path)) if os.path.exists(path) else {}
    nm, nq, voxel, kind, _ = bench.CONFIGS[cfg]
    doc[cfg] = {"kernel": " / ".join(sorted(shapes.get("first_search", []))) + " (first search of a scan, transform fused); bounded re-search: "
                            + " / ".join(sorted(shapes.get("hinted_search", []))),
                  "kernel_source_sha16": bench.knn_source_hash(), "date": datetime.date.today().isoformat(),
                  "launches_averaged": first.get("launches_averaged", 0),
                  "hbm_bytes_per_launch": first.get("hbm_bytes_per_launch"),
                  "method": "scripts/pmc_knn.sh: separate rocprofv3 --pmc passes (--kernel-trace only) over scripts/knn_loop.py; "
                            "bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950: FETCH_SIZE counts half of a wide coalesced read, "
                            "MI355X_MICROARCH.md HBM section; WRITE_SIZE uncalibrated); Infinity-Cache hits are included in these "
                            f"memory-side counters (the map of {nm} points x 16 B is resident there after warm-up when it fits its 256 MiB); "
                            f"config {cfg}: {kind} scan of {nq} points",
                  "first_search": first, "hinted_search": out.get("hinted_search", {})}
    json.dump(doc, open(path, "w"), indent=1)
    print(cfg, json.dumps({k: doc[cfg][k] for k in ("kernel", "kernel_source_sha16", "hbm_bytes_per_launch")}, indent=1))
    print(cfg, "instructions per wavefront", first.get("instructions_per_wavefront"))


if __name__ == "__main__":
    main()
