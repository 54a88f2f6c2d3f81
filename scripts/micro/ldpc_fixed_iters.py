#!/usr/bin/env python3
"""Cost of one belief-propagation iteration: pure-noise LLRs (no block ever converges), (1944,1296), B = 16384, exactly 20
iterations per block.  Prints block-iterations per second for both algorithms; used under scripts/collect_pmc.py to get
the instruction counts of the LDS-resident kernel (csrc/ldpc_resident.hip)."""
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
    n, B, iters = 1944, 16384, 20
    code = _device_code(p)
    llr = np.random.RandomState(1).randn(B, n) * 2.0
    dev = Dev(lib)
    d_llr = dev.put(llr)
    d_dec, d_out, d_it = dev.empty(B * n), dev.empty(B * n * 8), dev.empty(B * 4)
    tm = ctypes.c_void_p()
    lib.cpx_timer_create(ctypes.byref(tm))
    for alg, name in ((1, "MSA"), (0, "SPA")):
        if len(sys.argv) > 1 and name not in sys.argv[1:]:
            continue
        best = 1e9
        for rep in range(3):
            lib.cpx_timer_start(tm, None)
            _lib.check(lib.cpx_ldpc_bp_decode_batch_dev(code, d_llr, B, alg, iters, d_dec, d_out, d_it, None))
            lib.cpx_timer_stop(tm, None)
            v = ctypes.c_float()
            lib.cpx_timer_elapsed_ms(tm, ctypes.byref(v))
            best = min(best, v.value)
        its = dev.get(d_it, (B,), np.int32)
        print("%s %.3f ms, mean iterations %.2f -> %.1f M block-iterations/s  [%s]" % (
            name, best, its.mean(), its.sum() / best / 1e3, _lib.last_kernel()), flush=True)
    dev.free()


if __name__ == "__main__":
    main()
