"""Compile the HIP sources of libcommpy_amd.so for gfx950 (in-tree, no JIT cache).

    python -m commpy_amd.build [--force]

hipcc cross-compiles without a GPU.  Device code is built with -ffp-contract=off: the decoders
reproduce the reference's float64 arithmetic operation by operation, and fusing a*b+c into an FMA
would change the rounding of path metrics / state metrics.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
LIB = os.path.join(CSRC, "libcommpy_amd.so")
SOURCES = ["runtime.hip", "viterbi.hip", "viterbi_cw.hip", "bcjr.hip", "ldpc.hip", "ldpc_resident.hip", "demod.hip", "linksim.hip", "encoders.hip",
           "comm.hip"]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return "hipcc"


HEADERS = [os.path.join(CSRC, h) for h in ("cpx_internal.h", "cpx_math.h", "demod_dev.h", "ldpc_dev.h")] + [os.path.join(INCLUDE, "commpy_amd.h")]
OBJDIR = os.path.join(CSRC, "build")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wall", "-Wno-unused-result",
         "-I", INCLUDE, "-I", CSRC]
LINK_LIBS = ["-ldl"]            # librccl is dlopen()ed by comm.hip at the first communicator


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    mt = os.path.getmtime(target)
    return any(os.path.exists(d) and os.path.getmtime(d) > mt for d in deps)


def needs_build():
    return _stale(LIB, [os.path.join(CSRC, s) for s in SOURCES] + HEADERS)


def build_native(force=False, verbose=True):
    """Build libcommpy_amd.so; returns its path.  One object per translation unit (compiled in parallel, only the
    stale ones unless ``force``), then one link."""
    if not force and not needs_build():
        return LIB
    os.makedirs(OBJDIR, exist_ok=True)
    hipcc = _hipcc()
    jobs, objs = [], []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(OBJDIR, os.path.splitext(s)[0] + ".o")
        objs.append(obj)
        if force or _stale(obj, [src] + HEADERS):
            cmd = [hipcc] + FLAGS + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            jobs.append((s, subprocess.Popen(cmd)))
    failed = [s for s, pr in jobs if pr.wait() != 0]
    if failed:
        raise subprocess.CalledProcessError(1, "hipcc -c " + " ".join(failed))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + LINK_LIBS
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build_native(force="--force" in sys.argv)
    print(LIB)
