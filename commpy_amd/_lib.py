"""ctypes binding of libcommpy_amd.so (the HIP engine).  Declares exactly the symbols of
include/commpy_amd.h.  There is NO CPU fallback: if the library is missing or no HIP device is
usable, the decoders raise -- loudly -- instead of computing on the host.
"""
import ctypes
import os
from ctypes import (POINTER, c_char_p, c_double, c_float, c_int, c_int8, c_int32, c_int64, c_size_t,
                    c_uint8, c_uint64, c_void_p)

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# CPX_LIB_PATH: load another build of the engine (kernel experiments); the default is the in-tree library
LIB_PATH = os.environ.get("CPX_LIB_PATH") or os.path.join(_HERE, "csrc", "libcommpy_amd.so")

CPX_OK, CPX_EINVAL, CPX_EHIP, CPX_ENOMEM, CPX_ENODEV, CPX_ELIMIT = 0, -1, -2, -3, -4, -5

_dp = POINTER(c_double)
_i32p = POINTER(c_int32)
_u8p = POINTER(c_uint8)
_i8p = POINTER(c_int8)

# name -> (restype, argtypes); mirrors include/commpy_amd.h one to one
SYMBOLS = {
    "cpx_last_error": (c_char_p, []),
    "cpx_version": (c_int, []),
    "cpx_build_id": (c_char_p, []),
    "cpx_device_count": (c_int, [POINTER(c_int)]),
    "cpx_set_device": (c_int, [c_int]),
    "cpx_get_device": (c_int, [POINTER(c_int)]),
    "cpx_last_kernel": (c_int, [c_char_p, c_int]),
    "cpx_set_precision": (c_int, [c_char_p]),
    "cpx_get_precision": (c_int, []),
    "cpx_device_info": (c_int, [c_char_p, c_int, POINTER(c_int), POINTER(c_int64)]),
    "cpx_malloc": (c_int, [POINTER(c_void_p), c_size_t]),
    "cpx_free": (c_int, [c_void_p]),
    "cpx_memset": (c_int, [c_void_p, c_int, c_size_t]),
    "cpx_memcpy_h2d": (c_int, [c_void_p, c_void_p, c_size_t]),
    "cpx_memcpy_d2h": (c_int, [c_void_p, c_void_p, c_size_t]),
    "cpx_memcpy_h2d_async": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p]),
    "cpx_memcpy_d2h_async": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p]),
    "cpx_memcpy_d2d_async": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p]),
    "cpx_stream_create": (c_int, [POINTER(c_void_p)]),
    "cpx_stream_destroy": (c_int, [c_void_p]),
    "cpx_stream_sync": (c_int, [c_void_p]),
    "cpx_release_workspace": (c_int, []),
    "cpx_default_stream": (c_void_p, []),
    "cpx_timer_create": (c_int, [POINTER(c_void_p)]),
    "cpx_timer_start": (c_int, [c_void_p, c_void_p]),
    "cpx_timer_stop": (c_int, [c_void_p, c_void_p]),
    "cpx_timer_elapsed_ms": (c_int, [c_void_p, POINTER(c_float)]),
    "cpx_timer_destroy": (c_int, [c_void_p]),
    "cpx_sclk_probe_start": (c_int, [POINTER(c_void_p), c_double]),
    "cpx_sclk_probe_read": (c_int, [c_void_p, POINTER(c_double), POINTER(c_double)]),
    "cpx_sclk_probe_destroy": (c_int, [c_void_p]),
    "cpx_trellis_create": (c_int, [c_int, c_int, c_int, c_int, _i32p, _i32p, POINTER(c_void_p)]),
    "cpx_trellis_destroy": (c_int, [c_void_p]),
    "cpx_viterbi_decode_batch": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int64, c_int, c_int,
                                         c_void_p]),
    "cpx_viterbi_decode_batch_dev": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int64, c_int, c_int,
                                             c_void_p, c_void_p]),
    "cpx_viterbi_decode_batch_i64": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int64, c_int, c_int,
                                     c_void_p]),
    "cpx_viterbi_set_path": (c_int, [c_char_p]),
    "cpx_trellis_viterbi_spec_query": (c_int, [c_void_p, POINTER(c_int), POINTER(ctypes.c_uint), POINTER(ctypes.c_uint)]),
    "cpx_trellis_attach_viterbi_code": (c_int, [c_void_p, c_void_p, ctypes.c_size_t]),
    "cpx_trellis_detach_viterbi_code": (c_int, [c_void_p]),
    "cpx_trellis_has_viterbi_code": (c_int, [c_void_p]),
    "cpx_ldpc_set_path": (c_int, [c_char_p]),
    "cpx_demod_set_path": (c_int, [c_char_p]),
    "cpx_demod_hard_viterbi_batch": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int64, c_int,
                                             c_void_p]),
    "cpx_demod_hard_viterbi_batch_dev": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int64, c_int,
                                                 c_void_p, c_void_p]),
    "cpx_map_decode_batch": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_double, c_int,
                                     c_void_p, c_void_p]),
    "cpx_map_decode_batch_dev": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_double, c_int,
                                         c_void_p, c_void_p, c_void_p]),
    "cpx_turbo_decode_batch": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64,
                                       c_double, c_int, c_void_p]),
    "cpx_turbo_decode_batch_dev": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64,
                                           c_int64, c_double, c_int, c_void_p, c_void_p]),
    "cpx_ldpc_create": (c_int, [c_int, c_int, c_int64, _i32p, _i32p, POINTER(c_void_p)]),
    "cpx_ldpc_destroy": (c_int, [c_void_p]),
    "cpx_ldpc_blob_build": (c_int, [c_int, c_int, c_int64, _i32p, _i32p, c_void_p, c_size_t, POINTER(c_size_t)]),
    "cpx_ldpc_blob_info": (c_int, [c_void_p, c_size_t, POINTER(c_int), POINTER(c_int), POINTER(c_int64), POINTER(c_int),
                                   POINTER(c_int)]),
    "cpx_ldpc_create_from_blob": (c_int, [c_void_p, c_size_t, POINTER(c_void_p)]),
    "cpx_ldpc_bp_decode_batch": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "cpx_ldpc_bp_decode_batch_dev": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                             c_void_p]),
    "cpx_ldpc_bp_decode_batch_bm": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "cpx_ldpc_bp_decode_batch_bm_dev": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                                c_void_p]),
    "cpx_modem_create": (c_int, [c_void_p, c_int, POINTER(c_void_p)]),
    "cpx_modem_destroy": (c_int, [c_void_p]),
    "cpx_demod_soft": (c_int, [c_void_p, c_void_p, c_int64, c_double, c_void_p]),
    "cpx_demod_soft_dev": (c_int, [c_void_p, c_void_p, c_int64, c_double, c_void_p, c_void_p]),
    "cpx_demod_soft_scaled_dev": (c_int, [c_void_p, c_void_p, c_int64, c_double, c_double, c_void_p, c_void_p]),
    "cpx_demod_hard": (c_int, [c_void_p, c_void_p, c_int64, c_void_p]),
    "cpx_demod_hard_dev": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p]),
    "cpx_random_bits_dev": (c_int, [c_void_p, c_int64, c_uint64, c_uint64, c_void_p]),
    "cpx_conv_encode_batch_dev": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int, c_int, c_void_p, c_int64, c_void_p]),
    "cpx_gather_u8_dev": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_int64, c_void_p, c_void_p]),
    "cpx_gather_f64_dev": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_int64, c_void_p, c_void_p]),
    "cpx_modulate_dev": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p]),
    "cpx_awgn_dev": (c_int, [c_void_p, c_int64, c_double, c_double, c_uint64, c_uint64, c_void_p, c_void_p]),
    "cpx_scale_f64_dev": (c_int, [c_void_p, c_int64, c_double, c_void_p, c_void_p]),
    "cpx_bsc_dev": (c_int, [c_void_p, c_int64, c_double, c_uint64, c_uint64, c_void_p, c_void_p, c_void_p]),
    "cpx_bec_dev": (c_int, [c_void_p, c_int64, c_double, c_uint64, c_uint64, c_void_p, c_void_p, c_void_p]),
    "cpx_count_errors_dev": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int64, c_int64, c_void_p, c_void_p]),
    "cpx_link_front_create": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, POINTER(c_void_p)]),
    "cpx_link_front_destroy": (c_int, [c_void_p]),
    "cpx_link_front_run_dev": (c_int, [c_void_p, c_int64, c_double, c_double, c_double, c_double, c_uint64, c_uint64, c_uint64,
                                       c_void_p, c_void_p, c_void_p, c_void_p]),
    "cpx_turbo_encode_batch_dev": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_void_p,
                                           c_void_p, c_int64, c_int, c_void_p]),
    "cpx_ldpc_encoder_create": (c_int, [c_void_p, c_int64, c_int64, POINTER(c_void_p)]),
    "cpx_ldpc_encoder_destroy": (c_int, [c_void_p]),
    "cpx_ldpc_encode_batch_dev": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p]),
    "cpx_comm_unique_id": (c_int, [c_void_p]),
    "cpx_comm_init_rank": (c_int, [c_void_p, c_int, c_int, POINTER(c_void_p)]),
    "cpx_comm_init_all": (c_int, [POINTER(c_int), c_int, POINTER(c_void_p)]),
    "cpx_comm_info": (c_int, [c_void_p, POINTER(c_int), POINTER(c_int), POINTER(c_int)]),
    "cpx_comm_destroy": (c_int, [c_void_p]),
    "cpx_comm_allgather_u8": (c_int, [c_void_p, POINTER(c_void_p), POINTER(c_void_p), c_size_t, POINTER(c_void_p)]),
    "cpx_comm_allreduce_i64": (c_int, [c_void_p, POINTER(c_void_p), POINTER(c_void_p), c_size_t, c_int, POINTER(c_void_p)]),
    "cpx_comm_allreduce_f64": (c_int, [c_void_p, POINTER(c_void_p), POINTER(c_void_p), c_size_t, c_int, POINTER(c_void_p)]),
}


class EngineError(RuntimeError):
    """The HIP engine is missing or failed (no CPU fallback exists)."""


_lib = None


def load():
    """Load libcommpy_amd.so once; raises EngineError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise EngineError(
            "libcommpy_amd.so not found at %s -- build it with `python -m commpy_amd.build` "
            "(hipcc --offload-arch=gfx950). commpy_amd has no CPU fallback." % LIB_PATH)
    try:
        lib = ctypes.CDLL(LIB_PATH)
    except OSError as exc:  # pragma: no cover - depends on the host
        raise EngineError("cannot load %s: %s" % (LIB_PATH, exc)) from exc
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def last_error():
    msg = load().cpx_last_error()
    return msg.decode("utf-8", "replace") if msg else ""


def check(rc, value_error=ValueError):
    """Map an engine return code to the reference's exception types."""
    if rc == CPX_OK:
        return
    msg = last_error()
    if rc in (CPX_EINVAL, CPX_ELIMIT):
        raise value_error(msg)
    if rc == CPX_ENOMEM:
        raise MemoryError(msg)
    raise EngineError(msg or "libcommpy_amd error %d" % rc)


def device_count():
    n = c_int(0)
    load().cpx_device_count(ctypes.byref(n))
    return n.value


def require_device():
    if device_count() <= 0:
        raise EngineError("no HIP device available; commpy_amd computes on MI355X only (no CPU fallback)")


def ptr(arr):
    """void* of a C-contiguous NumPy array."""
    return arr.ctypes.data_as(c_void_p)


def as_f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def as_i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def current_device():
    d = c_int(0)
    check(load().cpx_get_device(ctypes.byref(d)))
    return d.value


def last_kernel():
    """Name of the kernel the last decoder call of this thread launched (cpx_last_kernel)."""
    buf = ctypes.create_string_buffer(400)
    check(load().cpx_last_kernel(buf, 400))
    return buf.value.decode()


def build_id():
    """{'full': sha16, 'viterbi': sha16}: digests of the sources the loaded library was compiled from (cpx_build_id)."""
    raw = load().cpx_build_id()
    txt = raw.decode() if raw else ""
    return dict(part.split(":", 1) for part in txt.split(";") if ":" in part)


def set_precision(mode):
    """'fp64-parity' (default; None resets to it) or 'fp32-fast' (float32 variants where they exist; not bit-exact)."""
    check(load().cpx_set_precision(None if mode is None else mode.encode()))


def get_precision():
    return "fp32-fast" if load().cpx_get_precision() else "fp64-parity"


def viterbi_last_path():
    """'fused' | 'cw2' | 'wave' | 'wide' | 'general' | 'fused+wave' ...: the Viterbi kernel path of the last call, from cpx_last_kernel."""
    k = last_kernel()
    parts = []
    if "viterbi_cw_fused_kernel" in k:
        parts.append("fused")
    if "viterbi_cw_acs_kernel" in k:
        parts.append("cw2")
    if "viterbi_wave_kernel" in k:
        parts.append("wave")
    if "viterbi_wide_kernel" in k:
        parts.append("wide")
    if "viterbi_generic_kernel" in k:
        parts.append("general")
    return "+".join(parts)


def viterbi_set_path(mode):
    """Force a Viterbi kernel path: None/'auto', 'wave', 'cw', 'cw!', 'cw2', 'cw2!', 'general' (tests and benchmarks)."""
    check(load().cpx_viterbi_set_path(None if mode is None else mode.encode()))


def demod_set_path(mode):
    """Soft-demodulator form: None/'auto' (two exponentials per axis for square QAM of 64 points and more, table-driven exp / log),
    'libm' (the same with the library's exp / log) or 'plain' (one exponential per level)."""
    check(load().cpx_demod_set_path(None if mode is None else mode.encode()))


def ldpc_set_path(mode):
    """Force an LDPC decoder path: None/'auto', 'tiled' (HBM-resident tiles), 'resident' (LDS-resident, strict), 'resident-log' (the same with
    the log-domain sum-product row instead of the ratio-domain kernel)."""
    check(load().cpx_ldpc_set_path(None if mode is None else mode.encode()))


class DeviceHandles:
    """Opaque engine handles of ONE host object (a Trellis, a Modem, an LDPC code ...), one per device.

    A handle owns small tables in the HBM of the device that was current when it was created, and the engine refuses it
    on any other device (CPX_EINVAL); so a host object that is used after ``cpx_set_device(other)`` -- the single-process
    multi-GPU driver of commpy_amd.parallel does exactly that -- gets a second handle there instead of a fault.
    """

    def __init__(self, create, destroy_name):
        self._create = create              # () -> c_void_p, on the current device
        self._destroy_name = destroy_name
        self._h = {}

    def get(self):
        dev = current_device()
        h = self._h.get(dev)
        if h is None:
            require_device()
            h = self._h[dev] = self._create()
        return h

    def drop(self):
        hs, self._h = self._h, {}
        for h in hs.values():
            try:
                getattr(load(), self._destroy_name)(h)
            except Exception:              # interpreter shutdown
                pass

    def __del__(self):
        self.drop()
