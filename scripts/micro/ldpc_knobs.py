#!/usr/bin/env python3
"""Experiment: LDS-resident LDPC kernel, workgroup size (CPX_LDPC_THREADS is read per call by csrc/ldpc_resident.hip).  (1944,1296), B = 32768, <= 50 iterations, Eb/N0 = 3 dB and 2.2 dB."""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "benchmarks"))
from commpy_amd import _lib  # noqa: E402
from bench_kernels import Dev  # noqa: E402


def main():
    from commpy_amd.channelcoding.ldpc import _device_code, get_ldpc_code_params
    lib = _lib.load()
    p = get_ldpc_code_params(os.path.join(ROOT, "commpy_amd/channelcoding/designs/ldpc/ieee80211n/1944.1296.txt"), True)
    n, B = 1944, 32768
    code = _device_code(p)
    rs = np.random.RandomState(31)
    for ebn0 in (3.0, 2.2):
        sigma = 1 / np.sqrt(10 ** (ebn0 / 10.0) * (2.0 / 3) * 2)
        llr = (2.0 * (1.0 + sigma * rs.randn(B, n)) / sigma ** 2)
        dev = Dev(lib)
        d_llr = dev.empty(llr.nbytes)
        d_dec, d_out, d_it = dev.empty(B * n), dev.empty(B * n * 8), dev.empty(B * 4)
        tm = ctypes.c_void_p()
        lib.cpx_timer_create(ctypes.byref(tm))
        for alg, name in ((1, "MSA"), (0, "SPA")):
            for knobs in sys.argv[1:] or ["", "T=1024", "T=768", "T=704", "T=512", "T=384", "T=256"]:
                os.environ.pop("CPX_LDPC_THREADS", None)
                os.environ.pop("CPX_LDPC_G", None)
                for kv in filter(None, knobs.split(",")):
                    k, v = kv.split("=")
                    os.environ["CPX_LDPC_THREADS" if k == "T" else "CPX_LDPC_G"] = v
                best = 1e9
                for rep in range(3):
                    _lib.check(lib.cpx_memcpy_h2d(d_llr, _lib.ptr(llr), llr.nbytes))
                    lib.cpx_timer_start(tm, None)
                    _lib.check(lib.cpx_ldpc_bp_decode_batch_bm_dev(code, d_llr, B, alg, 50, d_dec, d_out, d_it, None))
                    lib.cpx_timer_stop(tm, None)
                    v = ctypes.c_float()
                    lib.cpx_timer_elapsed_ms(tm, ctypes.byref(v))
                    best = min(best, v.value)
                print("%.1f dB %s %-12s %7.3f ms  %s" % (ebn0, name, knobs or "default", best, _lib.last_kernel()), flush=True)
        dev.free()


if __name__ == "__main__":
    main()
