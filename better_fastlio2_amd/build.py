"""Builds better_fastlio2_amd/lib/liblio_hip.so with hipcc for gfx950 (cross-compiles without a GPU).

    python -m better_fastlio2_amd.build [--force] [--verbose]

Flags: -O3 -ffp-contract=off (the fp32/fp64 decisions of the measurement path must not be fused,
see csrc/lio_math.h).  Objects are rebuilt only when a source or header is newer.
"""
from __future__ import annotations

import concurrent.futures as cf
import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
OBJ = os.path.join(PKG, "lib", "obj")
LIB = os.path.join(PKG, "lib", "liblio_hip.so")
ARCH = "gfx950"

SOURCES = ["lio_knn.hip", "lio_measure.hip", "lio_map.hip", "lio_mapadd.hip", "lio_scanprep.hip", "lio_sort.hip", "lio_rebuild.hip", "lio_compact.hip", "lio_gsort.hip", "lio_replicate.hip", "lio_ctx.hip", "lio_update.hip", "lio_scan.hip", "lio_pull.hip", "lio_pass.hip", "lio_capi_eskf.hip",
           "lio_esekf.cpp", "lio_imu.cpp"]
COMMON = ["-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wall", "-Wno-unused-function",
          "-Wno-unused-result", "-I", CSRC, "-I", os.path.join(ROOT, "include")]


def hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: liblio_hip.so cannot be built (there is no CPU fallback)")


def _newest_header() -> float:
    t = 0.0
    for d in (CSRC, os.path.join(ROOT, "include")):
        for f in os.listdir(d):
            if f.endswith((".h", ".hpp")):
                t = max(t, os.path.getmtime(os.path.join(d, f)))
    return t


def _compile(src: str, force: bool, verbose: bool) -> str:
    obj = os.path.join(OBJ, os.path.splitext(src)[0] + ".o")
    path = os.path.join(CSRC, src)
    if not force and os.path.exists(obj) and os.path.getmtime(obj) >= max(os.path.getmtime(path), _newest_header()):
        return obj
    if src.endswith(".cpp"):
        # host-only translation unit (the ESKF mirror): plain C++ through the same clang
        cmd = [hipcc(), "-x", "c++", "-c", path, "-o", obj] + COMMON
    else:
        cmd = [hipcc(), f"--offload-arch={ARCH}", "-x", "hip", "-c", path, "-o", obj] + COMMON
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed on {src}:\n{r.stdout}\n{r.stderr}")
    if verbose and r.stderr.strip():
        print(r.stderr)
    return obj


def build(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(OBJ, exist_ok=True)
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    with cf.ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        objs = list(ex.map(lambda s: _compile(s, force, verbose), srcs))
    if force or not os.path.exists(LIB) or any(os.path.getmtime(o) > os.path.getmtime(LIB) for o in objs):
        cmd = [hipcc(), f"--offload-arch={ARCH}", "--hip-link", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return LIB


def build_callback_bench() -> str:
    """examples/callback_bench.cpp -> lib/libcallback_bench.so: the zero-patch drop-in chain (the adaptor header behind the
    reference's callback signature) as a small host library that bench.py and the tests drive; plain g++ against the C ABI."""
    lib = build()
    src = os.path.join(ROOT, "examples", "callback_bench.cpp")
    out = os.path.join(PKG, "lib", "libcallback_bench.so")
    deps = [src, os.path.join(ROOT, "include", "lio_hip_adaptor.hpp"), os.path.join(ROOT, "include", "lio_hip.h"), lib]
    if os.path.exists(out) and os.path.getmtime(out) >= max(os.path.getmtime(d) for d in deps):
        return out
    cmd = ["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-Wall", "-I", os.path.join(ROOT, "include"),
           "-I", os.path.join(ROOT, "tests", "csrc", "adaptor_stubs"), src, "-o", out, "-L", os.path.dirname(lib), "-llio_hip",
           "-Wl,-rpath," + os.path.dirname(lib)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"g++ failed on callback_bench.cpp:\n{r.stdout}\n{r.stderr}")
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="--verbose" in sys.argv or "-v" in sys.argv))
    print(build_callback_bench())
