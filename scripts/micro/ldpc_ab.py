#!/usr/bin/env python3
"""A/B timing of the LDS-resident LDPC kernels for library variants (CPX_LIB_PATH selects the build): (1944,1296), B = 32768, block-major,
min-sum and sum-product at 2.2 dB / 3.0 dB (BPSK-like LLRs: a block takes 3 - 10 iterations, the per-block part matters) and on pure noise
with 50 iterations (the per-iteration part only).  One compact line per case: mean / min of 8 event-timed decodes after a warm-up.
    CPX_LIB_PATH=.../libX.so python scripts/micro/ldpc_ab.py <label>"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from commpy_amd import _lib  # noqa: E402
from benchmarks.other_configs import Dev, time_steps  # noqa: E402


def main():
    from commpy_amd.channelcoding.ldpc import _device_code, get_ldpc_code_params
    label = sys.argv[1] if len(sys.argv) > 1 else "?"
    lib = _lib.load()
    p = get_ldpc_code_params(os.path.join(ROOT, "commpy_amd/channelcoding/designs/ldpc/ieee80211n/1944.1296.txt"), True)
    n, B = 1944, 32768
    code = _device_code(p)
    rs = np.random.RandomState(31)
    cases = []
    for ebn0 in (2.2, 3.0):
        sigma = 1 / np.sqrt(10 ** (ebn0 / 10.0) * (2.0 / 3) * 2)
        cases.append(("%.1f dB" % ebn0, 2.0 * (1.0 + sigma * rs.randn(B, n)) / sigma ** 2))
    cases.append(("noise x50", rs.randn(B, n) * 2.0))
    dev = Dev(lib)
    d_dec, d_out, d_it = dev.empty(B * n), dev.empty(B * n * 8), dev.empty(B * 4)
    out = []
    for name, llr in cases:
        d_llr = dev.put(np.ascontiguousarray(llr))
        for alg, an in ((1, "MSA"), (0, "SPA")):
            ms = time_steps(lib, lambda: _lib.check(lib.cpx_ldpc_bp_decode_batch_bm_dev(code, d_llr, B, alg, 50, d_dec, d_out, d_it, None)), 8, 3)
            its = dev.get(d_it, (B,), np.int32)
            out.append("%s %s %.3f/%.3f (it %.1f)" % (an, name, float(np.mean(ms)), float(np.min(ms)), its.mean()))
    print(label, "|", " | ".join(out), "|", _lib.last_kernel()[:40])
    dev.free()


if __name__ == "__main__":
    main()
