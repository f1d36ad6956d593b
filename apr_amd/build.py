"""Build libapr_hip.so (the C-ABI HIP library) in-tree for gfx950 with hipcc.

`python -m apr_amd.build` or `apr_amd.build.build()`; rebuilds only when a source
is newer than the library.  The .so is git-ignored but travels with gpurun.
"""
from __future__ import annotations

import glob
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libapr_hip.so")
ARCH = "gfx950"


def _sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _deps():
    return _sources() + glob.glob(os.path.join(CSRC, "*.h")) + [
        os.path.join(HERE, "..", "include", "apr_hip.h")]


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(s) > t for s in _deps())


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not needs_build():
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: cannot build libapr_hip.so")
    os.makedirs(LIBDIR, exist_ok=True)
    objs = []
    jobs = []
    for src in _sources():
        obj = os.path.join(LIBDIR, os.path.basename(src).replace(".hip", ".o"))
        objs.append(obj)
        if (not force and os.path.exists(obj) and os.path.getmtime(obj) > os.path.getmtime(src)
                and all(os.path.getmtime(obj) > os.path.getmtime(h) for h in _deps() if h.endswith(".h"))):
            continue
        cmd = [hipcc, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden",
               "-Wall", "-Wno-unused-function", "-c", src, "-o", obj]
        if verbose:
            print("[apr_amd.build]", " ".join(cmd), flush=True)
        jobs.append((src, subprocess.Popen(cmd)))
    for src, p in jobs:
        if p.wait() != 0:
            raise RuntimeError(f"hipcc failed on {src}")
    cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB] + objs + [
        "-Wl,-rpath,/opt/rocm/lib"]
    if verbose:
        print("[apr_amd.build]", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
