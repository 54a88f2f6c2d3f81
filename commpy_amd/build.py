"""Compile the HIP sources of libcommpy_amd.so for gfx950 (in-tree, no JIT cache).

    python -m commpy_amd.build [--force]

hipcc cross-compiles without a GPU.  Device code is built with -ffp-contract=off: the decoders
reproduce the reference's float64 arithmetic operation by operation, and fusing a*b+c into an FMA
would change the rounding of path metrics / state metrics.
"""
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
LIB = os.path.join(CSRC, "libcommpy_amd.so")
SOURCES = ["runtime.hip", "viterbi.hip", "viterbi_cw.hip", "viterbi_generic.hip", "bcjr.hip", "bcjr_exact.hip", "ldpc.hip", "ldpc_resident.hip", "demod.hip", "linksim.hip", "encoders.hip",
           "comm.hip"]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return "hipcc"


HEADERS = [os.path.join(CSRC, h) for h in ("cpx_internal.h", "cpx_math.h", "demod_dev.h", "cpx_rng.h", "ldpc_dev.h", "viterbi_cw_asm.h")] + [os.path.join(INCLUDE, "commpy_amd.h")]
OBJDIR = os.path.join(CSRC, "build")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fvisibility=hidden", "-Wall", "-Wno-unused-result",
         "-I", INCLUDE, "-I", CSRC]
LINK_LIBS = ["-ldl"]            # librccl is dlopen()ed by comm.hip at the first communicator


def _digest(names):
    h = hashlib.sha256()
    for n in sorted(names):
        path = n if os.path.isabs(n) else os.path.join(CSRC, n)
        h.update(os.path.basename(path).encode() + b"\0")
        with open(path, "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


# what the Viterbi kernels are compiled from: bench.py only trusts committed rocprofv3 counters of the headline kernel when
# they were recorded with a library built from exactly these sources (cpx_build_id() of the .so == the id in the PMC file)
VITERBI_SOURCES = ["viterbi.hip", "viterbi_cw.hip", "viterbi_generic.hip", "viterbi_cw_asm.h", "cpx_math.h", "cpx_internal.h", "demod_dev.h"]


def source_build_id():
    """'full:<sha16 of every source and header>;viterbi:<sha16 of the Viterbi kernels' sources>' -- also compiled into the
    library (cpx_build_id), so that a measurement file can be tied to the code that produced it."""
    return "full:%s;viterbi:%s" % (_digest(SOURCES + HEADERS), _digest(VITERBI_SOURCES))


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
    build_id = source_build_id()
    id_file = os.path.join(OBJDIR, "build_id.txt")
    id_changed = not os.path.exists(id_file) or open(id_file).read() != build_id
    jobs, objs = [], []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(OBJDIR, os.path.splitext(s)[0] + ".o")
        objs.append(obj)
        if force or _stale(obj, [src] + HEADERS) or (s == "runtime.hip" and id_changed):
            cmd = [hipcc] + FLAGS + ["-c", src, "-o", obj]
            if s == "runtime.hip":
                cmd.insert(-4, '-DCPX_BUILD_ID="%s"' % build_id)      # cpx_build_id()
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
    with open(id_file, "w") as f:
        f.write(build_id)
    return LIB


if __name__ == "__main__":
    build_native(force="--force" in sys.argv)
    print(LIB)
