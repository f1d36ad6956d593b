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


CFLAGS = ["-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-Wall", "-Wno-unused-function"]


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
        cmd = [hipcc, f"--offload-arch={ARCH}"] + CFLAGS + ["-c", src, "-o", obj]
        if verbose:
            print("[apr_amd.build]", " ".join(cmd), flush=True)
        jobs.append((src, subprocess.Popen(cmd)))
    for src, p in jobs:
        if p.wait() != 0:
            raise RuntimeError(f"hipcc failed on {src}")
    # the assembly check of spconv_os.hip runs BEFORE the link: a failing check must leave no library behind that the
    # next build() (needs_build() == False) would hand out silently
    if any(os.path.basename(src) == "spconv_os.hip" for src, _ in jobs):
        try:
            check_os_pipeline(verbose)
        except BaseException:
            for stale in (LIB, os.path.join(LIBDIR, "spconv_os.o")):
                if os.path.exists(stale):
                    os.remove(stale)
            raise
    cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB] + objs + [
        "-Wl,-rpath,/opt/rocm/lib"]
    if verbose:
        print("[apr_amd.build]", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


COPY_OPS = ("v_mov_b32_e32", "v_mov_b32_e64", "v_mov_b64_e32", "v_mov_b64_e64", "v_pk_mov_b32", "v_swap_b32", "v_swap_b32_e32",
            "v_accvgpr_write_b32", "v_accvgpr_mov_b32")
SPILL_PREFIXES = ("scratch_store", "buffer_store")


def check_os_pipeline(verbose: bool = True) -> None:
    """spconv_os.hip keeps gathered rows in flight in registers across the item loop's back edge (inline-asm loads +
    counted s_waitcnt).  A register copy of such a value made by the compiler BEFORE its wait would read stale data, and
    nothing at run time would say so: compile the file to assembly (the flags of the real compile) and refuse the build if
    any move / swap / spill instruction inside k_os_conv's item loop reads a register that an inline-asm
    global_load_dwordx4 writes (v_mov / v_pk_mov / v_swap / accvgpr writes, scratch_ / buffer_ stores = spills).  The
    run-time guard is tests/test_spconv_gpu.py::test_os_conv_debug_pipeline_matches_production."""
    import re
    import tempfile
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    src = os.path.join(CSRC, "spconv_os.hip")
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "os.s")
        subprocess.check_call([hipcc, f"--offload-arch={ARCH}"] + CFLAGS + ["-S", "--cuda-device-only",
                               "-Wno-unused-command-line-argument", "-o", out, src], stderr=subprocess.DEVNULL)
        text = open(out).read().splitlines()

    def regs(tok):
        m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
        if m:
            return set(range(int(m.group(1)), int(m.group(2)) + 1))
        m = re.fullmatch(r"v(\d+)", tok)
        return {int(m.group(1))} if m else set()

    # per kernel: the lines of the item loop = from the first counted wait of the pipeline to the last one
    bodies, kernel = {}, None
    for ln in text:
        m = re.match(r"^(_Z\S*k_os_conv\S*):", ln)
        if m:
            kernel = m.group(1)
            bodies[kernel] = []
        elif ln.strip().startswith(".Lfunc_end"):
            kernel = None
        elif kernel is not None:
            bodies[kernel].append(ln.strip())
    loaded, bad = {}, []
    for k, lines in bodies.items():
        marks = [i for i, t in enumerate(lines) if t.startswith("s_waitcnt vmcnt(5)") or t.startswith("s_waitcnt vmcnt(9)")]
        loaded[k] = set()
        if not marks:
            continue
        body = lines[marks[0]:marks[-1] + 1]
        in_asm, moves = False, []
        for t in body:
            if "#ASMSTART" in t:
                in_asm = True
            elif "#ASMEND" in t:
                in_asm = False
            ops = t.replace(",", " ").split()
            if not ops:
                continue
            if in_asm and ops[0] == "global_load_dwordx4":
                loaded[k] |= regs(ops[1])
            elif not in_asm and (ops[0] in COPY_OPS or ops[0].startswith(SPILL_PREFIXES)):
                moves.append((ops, t))
        for ops, t in moves:
            src_regs = set()
            for tok in (ops[1:] if ops[0].startswith(SPILL_PREFIXES) or ops[0].startswith("v_swap") else ops[2:]):
                src_regs |= regs(tok)
            if src_regs & loaded[k]:
                bad.append(f"{t}    [{k[:48]}]")
    if not loaded or not all(loaded.values()):
        raise RuntimeError("check_os_pipeline: found no inline-asm row loads in k_os_conv (the check is out of date)")
    if bad:
        raise RuntimeError("spconv_os.hip: the compiler copies registers of the in-flight row pipeline:\n  " +
                           "\n  ".join(bad[:8]))
    if verbose:
        print(f"[apr_amd.build] spconv_os pipeline check: {sum(len(v) for v in loaded.values())} row registers, "
              "no copies", flush=True)


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
