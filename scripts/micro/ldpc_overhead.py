#!/usr/bin/env python3
"""Per-block overhead of the LDS-resident LDPC kernels: pure-noise LLRs (no block converges), (1944,1296), B = 32768, block-major
outputs, exactly n iterations for n = 1, 2, 4, 8, 16 -- time = B x (overhead + n x per-iteration cost); prints the fitted pair.
(round 5: at the SNRs where a link operates a block takes 2 - 6 iterations and the per-block part dominates)"""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from commpy_amd import _lib  # noqa: E402
from benchmarks.other_configs import Dev, time_steps  # noqa: E402


def main():
    from commpy_amd.channelcoding.ldpc import _device_code, get_ldpc_code_params
    lib = _lib.load()
    p = get_ldpc_code_params(os.path.join(ROOT, "commpy_amd/channelcoding/designs/ldpc/ieee80211n/1944.1296.txt"), True)
    n, B = 1944, 32768
    code = _device_code(p)
    llr = np.random.RandomState(1).randn(B, n) * 2.0
    dev = Dev(lib)
    d_llr = dev.put(llr)
    d_dec, d_out, d_it = dev.empty(B * n), dev.empty(B * n * 8), dev.empty(B * 4)
    for alg, name in ((1, "MSA"), (0, "SPA")):
        xs, ys = [], []
        for iters in (1, 2, 4, 8, 16):
            ms = time_steps(lib, lambda: _lib.check(lib.cpx_ldpc_bp_decode_batch_bm_dev(code, d_llr, B, alg, iters, d_dec, d_out, d_it, None)), 5, 2)
            xs.append(iters); ys.append(float(np.median(ms)))
        a, b0 = np.polyfit(xs, ys, 1)
        print("%s: %s ms for %s iterations -> per iteration %.4f ms, per-block part %.4f ms (= %.1f iterations)  [%s]" % (
            name, " ".join("%.3f" % y for y in ys), xs, a, b0, b0 / a, _lib.last_kernel()), flush=True)
    dev.free()


if __name__ == "__main__":
    main()
