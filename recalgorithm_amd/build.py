"""Build librecalgo_hip.so (the C-ABI HIP library) in-tree with hipcc for gfx950.

    python -m recalgorithm_amd.build [--force]

hipcc cross-compiles without a GPU.  Each .hip translation unit is compiled to an object
(only when stale), then linked into `recalgorithm_amd/librecalgo_hip.so`, which is
git-ignored but travels to the GPU box with the gpurun snapshot.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
OBJDIR = os.path.join(HERE, "build")
LIB = os.path.join(HERE, "librecalgo_hip.so")

ARCH = "gfx950"
CXXFLAGS = [
    f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics",
    "-fvisibility=hidden", "-Wall", "-Wno-unused-function", f"-I{INCLUDE}",
]


def _hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (need ROCm with gfx950 support)")


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _deps_mtime() -> float:
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hdrs += [os.path.join(INCLUDE, f) for f in os.listdir(INCLUDE) if f.endswith(".h")]
    return max(os.path.getmtime(h) for h in hdrs)


def build(force: bool = False, verbose: bool = True) -> str:
    hipcc = _hipcc()
    os.makedirs(OBJDIR, exist_ok=True)
    hdr_m = _deps_mtime()
    jobs = []
    objs = []
    for src in sources():
        obj = os.path.join(OBJDIR, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        stale = force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hdr_m)
        if stale:
            jobs.append((src, obj))

    def compile_one(job):
        src, obj = job
        cmd = [hipcc, *CXXFLAGS, "-c", src, "-o", obj]
        if verbose:
            print("[recalgo build]", " ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)

    if jobs:
        with ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
            list(ex.map(compile_one, jobs))
    stale_lib = not os.path.exists(LIB) or any(os.path.getmtime(o) > os.path.getmtime(LIB) for o in objs)
    if jobs or stale_lib or force:
        cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", *objs, "-o", LIB]
        if verbose:
            print("[recalgo build]", " ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    return LIB


HOST_SRC = os.path.join(HERE, "csrc_host", "tfrecord_reader.cpp")
HOST_LIB = os.path.join(HERE, "librecalgo_host.so")


def build_host(force: bool = False, verbose: bool = True) -> str:
    """librecalgo_host.so: the native TFRecord / Example / vocabulary plumbing (g++, no GPU code)."""
    hdr = os.path.join(INCLUDE, "recalgo_host.h")
    stale = force or not os.path.exists(HOST_LIB) or \
        os.path.getmtime(HOST_LIB) < max(os.path.getmtime(HOST_SRC), os.path.getmtime(hdr))
    if stale:
        cxx = shutil.which("g++") or "g++"
        cmd = [cxx, "-O3", "-std=c++17", "-fPIC", "-shared", "-pthread", "-fvisibility=hidden", "-Wall", f"-I{INCLUDE}",
               HOST_SRC, "-o", HOST_LIB]
        if verbose:
            print("[recalgo build]", " ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    return HOST_LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
    print(build_host(force="--force" in sys.argv))
