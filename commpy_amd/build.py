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
SOURCES = ["runtime.hip", "viterbi.hip", "viterbi_cw.hip", "bcjr.hip", "ldpc.hip", "demod.hip", "linksim.hip", "encoders.hip"]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return "hipcc"


def needs_build():
    if not os.path.exists(LIB):
        return True
    mt = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + [os.path.join(CSRC, "cpx_internal.h"), os.path.join(CSRC, "cpx_math.h"),
                                                       os.path.join(INCLUDE, "commpy_amd.h")]
    return any(os.path.exists(d) and os.path.getmtime(d) > mt for d in deps)


def build_native(force=False, verbose=True):
    """Build libcommpy_amd.so; returns its path."""
    if not force and not needs_build():
        return LIB
    srcs = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
           "-Wall", "-Wno-unused-result", "-I", INCLUDE, "-I", CSRC, "-o", LIB] + srcs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build_native(force="--force" in sys.argv)
    print(LIB)
