"""Per-pair code objects for the codeword-per-lane Viterbi kernels (round 6; no reference counterpart).

The engine ships the fused kernel of ``csrc/viterbi_cw.hip`` compiled for the standard generator pairs; every other rate-1/2
code of full constraint length runs a table-driven flavour of the same kernel that is ~15 % slower.  ``viterbi_cw.hip`` can also
be compiled for ONE pair (``-DCPX_VIT_SPEC_LG / _G0 / _G1``; the pair becomes template arguments like the built-in ones) into a
small device-only code object.  This module drives that compile -- ``hipcc`` as a subprocess, ~20 s, no GPU needed --, caches the
result by (generators, digest of the Viterbi sources) and hands it to ``cpx_trellis_attach_viterbi_code``.  Nothing here
computes: without ``hipcc`` (or with a failing compile) the trellis simply keeps the table-driven kernel.

    from commpy_amd.channelcoding import Trellis
    tr = Trellis(np.array([6]), np.array([[0o135, 0o147]]))
    tr.specialize()            # True: a code object is attached; decoded bits are identical either way

``CPX_VITERBI_JIT=1`` in the environment makes every ``Trellis`` do this on first use; ``CPX_JIT_CACHE`` names the cache directory
(default: ``commpy_amd/csrc/build/jit`` next to the library, then ``~/.cache/commpy_amd``).
"""
import ctypes
import os
import subprocess
import tempfile

from commpy_amd import _lib
from commpy_amd import build as _build

__all__ = ['viterbi_code_object', 'specialize_trellis', 'has_code_object', 'compile_command', 'kernel_symbol']

FLAGS = ["--offload-arch=gfx950", "--cuda-device-only", "--no-gpu-bundle-output", "-O3", "-std=c++17", "-ffp-contract=off"]


def _cache_dirs():
    env = os.environ.get("CPX_JIT_CACHE")
    if env:
        return [env]
    return [os.path.join(_build.OBJDIR, "jit"), os.path.join(os.path.expanduser("~"), ".cache", "commpy_amd")]


def _viterbi_digest():
    return _build.source_build_id().split("viterbi:")[1]


def compile_command(lg, g0, g1, out):
    """The hipcc command line of one pair's code object (generators in the kernel template's convention: bit ``lg`` taps the input)."""
    return [_build._hipcc()] + FLAGS + ["-DCPX_VIT_SPEC_LG=%d" % lg, "-DCPX_VIT_SPEC_G0=%du" % g0, "-DCPX_VIT_SPEC_G1=%du" % g1,
                                        "-I", _build.INCLUDE, "-I", _build.CSRC, "-c", os.path.join(_build.CSRC, "viterbi_cw.hip"),
                                        "-o", out]


def kernel_symbol(lg, g0, g1, decoding_type, runtime_hops):
    """Mangled name of one of the six kernels of a pair's code object (what cpx_trellis_attach_viterbi_code looks up)."""
    ring = 32 if lg == 6 else (16 if 5 * lg - 1 <= 16 else 32)
    return ("_ZN12_GLOBAL__N_123viterbi_cw_fused_kernelILi%dELj%dELj%dELi%dELi%dELb%dEdLi%dELb%dEEEvNS_8CwParamsE"
            % (lg, g0, g1, decoding_type, 5 * lg - 2, int(bool(runtime_hops)), ring, 1 if lg == 6 else 0))


def viterbi_code_object(lg, g0, g1, timeout=600):
    """Bytes of the code object of pair (g0, g1), from the cache or freshly compiled; ``None`` when it cannot be built here
    (no hipcc, compile error, no writable cache) -- the reason is in ``viterbi_code_object.last_error``."""
    viterbi_code_object.last_error = None
    name = "vit_%s_%d_%o_%o.co" % (_viterbi_digest(), lg, g0, g1)
    for d in _cache_dirs():
        path = os.path.join(d, name)
        if os.path.exists(path) and os.path.getsize(path) > 64:
            with open(path, "rb") as f:
                return f.read()
    for d in _cache_dirs():
        try:
            os.makedirs(d, exist_ok=True)
            fd, tmp = tempfile.mkstemp(suffix=".co", dir=d)
            os.close(fd)
        except OSError as exc:
            viterbi_code_object.last_error = "cache directory %s: %s" % (d, exc)
            continue
        try:
            res = subprocess.run(compile_command(lg, g0, g1, tmp), capture_output=True, text=True, timeout=timeout)
            if res.returncode != 0 or os.path.getsize(tmp) <= 64:
                viterbi_code_object.last_error = "hipcc failed (%d): %s" % (res.returncode, res.stderr[-400:])
                return None
            os.replace(tmp, os.path.join(d, name))                  # atomic: concurrent processes see a whole file or none
            tmp = None
            with open(os.path.join(d, name), "rb") as f:
                return f.read()
        except (OSError, subprocess.TimeoutExpired) as exc:
            viterbi_code_object.last_error = "%s: %s" % (type(exc).__name__, exc)
            return None
        finally:
            if tmp and os.path.exists(tmp):
                os.unlink(tmp)
    return None


viterbi_code_object.last_error = None


def specialize_trellis(handle):
    """Attach the pair's code object to the ``cpx_trellis*`` ``handle`` if the library says it would gain from one.
    Returns True when a code object is attached afterwards, False when the trellis keeps the kernel it had (built-in pair, another
    structure, no compiler); raises only for a refused image (ValueError: sources and cache out of step)."""
    lib = _lib.load()
    lg, g0, g1 = ctypes.c_int(0), ctypes.c_uint(0), ctypes.c_uint(0)
    _lib.check(lib.cpx_trellis_viterbi_spec_query(handle, ctypes.byref(lg), ctypes.byref(g0), ctypes.byref(g1)))
    if lg.value == 0:
        return False
    image = viterbi_code_object(lg.value, g0.value, g1.value)
    if image is None:
        return False
    buf = ctypes.create_string_buffer(image, len(image))
    _lib.check(lib.cpx_trellis_attach_viterbi_code(handle, buf, len(image)))
    return True


def has_code_object(handle):
    """True when the ``cpx_trellis*`` already launches a per-pair code object (the query then reports nothing to gain, and the
    trellis is of the table-driven family)."""
    return bool(_lib.load().cpx_trellis_has_viterbi_code(handle))
